// kz_wavefront.h — the wavefront form of PathMisIntegrator::Li (integrator.cpp:195-338) for CDNA4.
//
// The megakernel keeps one lane on one path from camera to termination, so the lanes of a wave sit in four
// different inlined copies of the traversal loop and in shading code at the same time: rocprof shows ~13 % of
// lanes active per VALU instruction and 3 waves/SIMD (profiles/r01a_megakernel). Here the path is cut at every
// ray query into stages that each run ONE kind of work over a compacted queue of path slots:
//
//   generate                      renderSample up to the camera ray                       (renderer.cpp:20-33)
//   extend      (closest hit)     Accel::rayIntersect traversal half                      (accel.cpp:63-110)
//   primary-fix + extend-keep     the invisible-light walk-through of the first hit (H6)   (integrator.cpp:214-219)
//   shade(k)                      post-intersection, emitter hit, roulette, light sampling with BSDF eval/pdf and the
//                                 MIS weight, BSDF sampling -> shadow ray + pending radiance, next ray
//   shadow(k)                     the occlusion test; adds the pending radiance when unoccluded (integrator.cpp:257-295)
//   final                         background on a miss after the last bounce               (integrator.cpp:315-318)
//
// Path state lives in HBM as SoA (float4 per field, one slot per (pixel,sample) item of the pass: 33.5 M slots =
// ~5 GB of 288 GB), so every stage reads and writes coalesced 16-B records. Queues hold slot indices; kernels are
// persistent (grid-stride over 256-item chunks, count read from device memory: no host round trip per stage) and
// compact their survivors through an LDS staging buffer with one global atomic per ~2 K items.
// The arithmetic per path and the order of its radiance additions are those of the megakernel, so both pipelines
// produce bit-identical sample records.
#pragma once
#include "kz_state.h"
#include "kz_devfn.h"

#ifndef KZ_WF_QCAP
#define KZ_WF_QCAP 960             // LDS staging entries per output queue per workgroup (960: four shade workgroups fit the CU's 160 KB)
#endif
#define KZ_WF_ROUNDS (KZ_WF_QCAP / KZ_BLOCK)   // rounds of one entry per thread the staging buffer takes

// ---- workgroup-level compaction: append `val` for lanes with pred into an LDS buffer, flush to the global queue ----
struct WfAppender {
    uint32_t *s_buf; uint32_t *s_n; uint32_t *s_gb; uint32_t *gQueue; uint32_t *gCount;
    __device__ __forceinline__ void push(bool pred, uint32_t val) {
        const unsigned long long m = __ballot(pred);
        const int lane = threadIdx.x & 63;
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(s_n, (uint32_t)__popcll(m));
        base = __shfl(base, 0, 64);
        if (pred) s_buf[base + __popcll(m & ((1ull << lane) - 1ull))] = val;
    }
    // all threads of the workgroup call this; flushes when another 256-item chunk might not fit (or always, if force)
    __device__ __forceinline__ void maybeFlush(bool force) {
        __syncthreads();
        const uint32_t n = *s_n;
        if (force ? (n > 0) : (n > KZ_WF_QCAP - KZ_BLOCK)) {
            if (threadIdx.x == 0) *s_gb = atomicAdd(gCount, n);
            __syncthreads();
            const uint32_t gb = *s_gb;
            for (uint32_t i = threadIdx.x; i < n; i += KZ_BLOCK) gQueue[gb + i] = s_buf[i];
            __syncthreads();
            if (threadIdx.x == 0) *s_n = 0;
        }
        __syncthreads();
    }
};

__device__ __forceinline__ void wfStatsFlush(unsigned long long *stats, const Counters &cn, uint32_t samples) {
    unsigned long long v[7] = {samples, cn.rays, cn.nodes, cn.tris, cn.hits, cn.lsamples, cn.dropped};
    for (int k = 0; k < 7; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&stats[k], x);
    }
}

// dimPmj: the dimension counter of a pmj02bn path at this point of this bounce (wfPmjDim). That sampler has no other state, so its paths
// neither load nor store a sampler record (16 B each way per shaded hit).
__device__ __forceinline__ uint32_t wfPmjDim(const KzParams &P, int iter) {
    // generate leaves the counter at 4 (max(2, 0), then the aperture's 2-D draw); bounce j draws the roulette sample (j >= 3), the light pick,
    // three light-sample dimensions when there are lights, and 2 + 1 for the BSDF sample
    return 4u + (uint32_t)iter * (4u + (P.nLights > 0 ? 3u : 0u)) + (iter > 3 ? (uint32_t)(iter - 3) : 0u);
}
__device__ __forceinline__ void wfLoadSampler(const KzParams &P, const KzWf &W, uint32_t slot, int px, int py, uint32_t sampleIndex, Sampler &s, uint32_t dimPmj) {
    s.type = P.samplerType; s.px = px; s.py = py; s.idx = sampleIndex;
    if (P.samplerType == KZ_SAMPLER_PMJ02BN) {
        s.state = 0ull; s.inc = 0ull; s.dim = dimPmj; s.hp = hashPixelBlock(px, py);
        return;
    }
    const uint4 v = W.smp[slot];
    s.state = (uint64_t)v.x | ((uint64_t)v.y << 32);
    // Every path of one shade launch is at the same depth and has drawn the same number of dimensions (the draws of a bounce do not depend on
    // the path: roulette from depth 3, pick, 3 light draws when there are lights, 2-D + 1-D for the BSDF), so the dimension counter is
    // wave-uniform: taken from the first active lane it lives in an SGPR and the dimension/seed block of Hash(p, dim, seed) (two 64-bit
    // multiplies per draw) and the blue-noise texture index are computed on the scalar unit. The megakernel keeps the per-lane counter.
    s.dim = (uint32_t)__builtin_amdgcn_readfirstlane((int)v.z);
    s.inc = (P.samplerType == KZ_SAMPLER_PMJ02BN) ? 0ull : ((hashPixelSeed(px, py, P.seed) << 1u) | 1u);      // pcg32 stream id: a function of the pixel
    s.hp = (P.samplerType != KZ_SAMPLER_INDEPENDENT) ? hashPixelBlock(px, py) : 0ull;
}
__device__ __forceinline__ void wfStoreSampler(const KzParams &P, const KzWf &W, uint32_t slot, const Sampler &s) {
    uint4 v;
    if (P.samplerType == KZ_SAMPLER_PMJ02BN) return;                // (the dimension counter is a function of the bounce: wfPmjDim)
    v.x = (uint32_t)s.state; v.y = (uint32_t)(s.state >> 32); v.z = s.dim; v.w = 0;
    W.smp[slot] = v;
}

// ---- generate: renderSample up to the camera ray (renderer.cpp:20-33) ------------------------------------------------
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_generate(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ pixList,
                                                           uint32_t nItems, uint32_t S, uint32_t sampleBegin) {
    const uint32_t item = blockIdx.x * KZ_BLOCK + threadIdx.x;
    if (item >= nItems) return;
    // With a multiple of 64 samples per pixel (every default) the wave is ONE pixel: its index comes from the first lane's item, so the division, the
    // pixel-list entry, the pixel's share of the sampler hash and the blue-noise shifts of the aperture draw are scalar work instead of 64 copies of it.
    uint32_t pl, pxy;
    if ((S & 63u) == 0) { pl = (uint32_t)__builtin_amdgcn_readfirstlane((int)item) / S; pxy = (uint32_t)__builtin_amdgcn_readfirstlane((int)pixList[pl]); }
    else { pl = item / S; pxy = pixList[pl]; }
    const uint32_t so = item - pl * S;
    const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
    Sampler smp; smp.type = P.samplerType;
    smp.generateSample(P, T, px, py, sampleBegin + so);
    float jx, jy; smp.nextPixel2D(P, T, jx, jy);
    const float sx = (float)px + jx, sy = (float)py + jy;
    float ax, ay; smp.next2D(P, T, ax, ay);
    V3 ro, rd; float mint, maxt;
    cameraRay(P, sx, sy, ax, ay, ro, rd, mint, maxt);
    kzStoreStream(&W.rayA[item], make_float4(ro.x, ro.y, ro.z, mint));
    kzStoreStream(&W.rayB[item], make_float4(rd.x, rd.y, rd.z, maxt));
    // (throughput = 1, eta = 1, bsdfPdf / accumulatedRoughness / discrete = 0 are what shade(0) assumes for a camera path: not stored)
    wfStoreSampler(P, W, item, smp);
    W.outJx[item] = jx; W.outJy[item] = jy; W.outR[item] = 0.f; W.outG[item] = 0.f; W.outB[item] = 0.f;
}


// ---- primary fix: first hit on a light with lightPrimaryVisibility == false -> continuation ray (integrator.cpp:214-219)
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_primary_fix(KzParams P, KzDevTables T, KzWf W, uint32_t nItems, uint32_t *__restrict__ outQueue,
                                                              uint32_t *__restrict__ outCount) {
    __shared__ uint32_t s_buf[KZ_WF_QCAP]; __shared__ uint32_t s_n, s_gb;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    WfAppender ap = {s_buf, &s_n, &s_gb, outQueue, outCount};
    for (uint32_t base = blockIdx.x * KZ_BLOCK; base < nItems; base += gridDim.x * KZ_BLOCK) {
        const uint32_t slot = base + threadIdx.x;
        bool need = false;
        if (slot < nItems) {
            const float4 h = kzLoadStream(&W.hit[slot]);
            if (h.x < KZ_INF) {
                RawHit rh; rh.t = h.x; rh.u = h.y; rh.v = h.z; rh.tri = 0; rh.gid = __float_as_uint(h.w);
                const int li = lightOfGid(T, rh.gid);
                if (li >= 0 && !T.lights[li].primaryVisibility) {
                    Its its; postIntersect<false>(T, rh, its);
                    const float4 b = kzLoadStream(&W.rayB[slot]);
                    const V3 rd = mk(b.x, b.y, b.z);
                    const V3 no = its.p + P.traceBias * rd;
                    kzStoreStream(&W.shA[slot], make_float4(no.x, no.y, no.z, KZ_INF));            // Ray3f(o, d): mint = Epsilon, maxt = inf
                    kzStoreStream(&W.shB[slot], make_float4(rd.x, rd.y, rd.z, KZ_EPSILON));
                    need = true;
                }
            }
        }
        ap.push(need, slot);
        ap.maybeFlush(false);
    }
    ap.maybeFlush(true);
}

// ---- shade(iter): everything between two ray queries of Li ---------------------------------------------------------------
// Waves per SIMD the shade kernel is compiled for. The lean variant (constant diffuse / kiss rows) at 4 = 128 VGPRs (18 spilled to
// scratch) + 40.5 KB LDS: alone it is 4 % slower than at 3 waves / 150 VGPRs, but with two passes in flight the smaller workgroups get
// onto the CUs between the other pass's persistent traversal workgroups: +1.7 % on C4, +2.1 % on C3 (same gpurun call, r02e). The EXT
// variant (textures, other BSDF models) needs 168 VGPRs and 48.7 KB: 3.
#ifndef KZ_SHADE_WAVES
#define KZ_SHADE_WAVES 4
#endif
// development build only (-DKZ_SHADESTAT): wall-clock cycles (s_memtime, per wave, stalls included) each section of the shade kernels takes,
// summed over the waves into stats[8..23]; the shares say where the waves' time goes, whatever bounds it
struct KzSst {
#ifdef KZ_SHADESTAT
    unsigned long long t = __builtin_amdgcn_s_memtime(), acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void mark(int k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc[k] += t_ - t; t = t_; }
    // the same after every outstanding load has returned: the wait is charged to the section that issued the loads
    __device__ __forceinline__ void markw(int k) { __builtin_amdgcn_s_waitcnt(0); mark(k); }
    __device__ __forceinline__ void flush(unsigned long long *stats) { if ((threadIdx.x & 63) == 0) for (int k = 0; k < 14; ++k) atomicAdd(stats + 8 + k, acc[k]); }
#else
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void markw(int) {}
    __device__ __forceinline__ void flush(unsigned long long *) {}
#endif
};

// Pass A of shade(iter) for one queue entry: hit record -> intersection; a miss, an emitter hit and a one-sided BSDF seen from behind end the
// path here (integrator.cpp:210-231, 315-327), the Russian roulette of integrator.cpp:237-244 is played. True = the path goes on to the light
// sample and the BSDF sample (wfShadeSurvivor); `its` is then complete. `compact`: see kz_wf_shade.
template <bool STATS, int EXT>
__device__ __forceinline__ bool wfClassify(const KzParams &P, const KzDevTables &T, const KzWf &W, const uint32_t *__restrict__ pixList, uint32_t S, uint32_t sampleBegin,
                                           int iter, bool compact, uint32_t slot, Its &its, uint32_t &modelKey, Counters &cn, KzSst &sst) {
    bool survivor = false;
    const float4 h = kzLoadStream(&W.hit[slot]);
    const float4 rb = kzLoadStream(&W.rayB[slot]);
    const V3 rd = mk(rb.x, rb.y, rb.z);
#ifdef KZ_SHADESTAT
    sst.markw(10);                                  // pass A: queue entry, hit record, ray
    if (h.x < KZ_INF) { const float4 *sp_ = reinterpret_cast<const float4 *>(T.shade + __float_as_uint(h.w)); const float4 a_ = sp_[0], b_ = sp_[6]; asm volatile("" :: "v"(a_.x), "v"(b_.x)); }
    sst.markw(11);                                  // pass A: the shading record arrives
#endif
    if (!(h.x < KZ_INF)) {
        // miss: black for the primary ray (H5), background after a bounce (integrator.cpp:315-318)
        if (iter > 0 && P.bgPresent) {
            const float4 th = kzLoadStream(&W.thr[slot]);
            const V3 c = mk(th.x, th.y, th.z) * backgroundRadiance(P, T, rd);
            unsafeAtomicAdd(W.outR + slot, c.x); unsafeAtomicAdd(W.outG + slot, c.y); unsafeAtomicAdd(W.outB + slot, c.z);      // (returnless: no wait, one writer per slot)
        }
        return false;
    }
    RawHit rh; rh.t = h.x; rh.u = h.y; rh.v = h.z; rh.tri = 0; rh.gid = __float_as_uint(h.w);
    postIntersect<false>(T, rh, its); if (STATS) cn.hits++;
    sst.markw(12);                                  // pass A: postIntersect
    if (its.light >= 0) {                                                         // integrator.cpp:226-231, 322-327
        const KzLightRow &lr = T.lights[its.light];
        const float4 ra = kzLoadStream(&W.rayA[slot]);
        const float4 th = iter == 0 ? make_float4(1.f, 1.f, 1.f, 1.f) : kzLoadStream(&W.thr[slot]);
        const float4 mi = iter == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : (compact ? make_float4(th.w, 0.f, 0.f, 0.f) : kzLoadStream(&W.misc[slot]));
        const V3 ro = mk(ra.x, ra.y, ra.z);
        const V3 wi = normalized(its.p - ro);
        float bsdfWeight = 1.f;
        if (iter > 0 && mi.z == 0.f) bsdfWeight = powerHeuristic(mi.x, lightPdfSolidAngle(lr.normalization, its.sh.n, wi, its.p, ro));   // mi.z: EDiscrete (integrator.cpp:329-331)
        if (dot(its.sh.n, -wi) > 0.f) {
            const V3 c = (bsdfWeight * mk(th.x, th.y, th.z)) * mk(lr.radiance[0], lr.radiance[1], lr.radiance[2]);
            unsafeAtomicAdd(W.outR + slot, c.x); unsafeAtomicAdd(W.outG + slot, c.y); unsafeAtomicAdd(W.outB + slot, c.z);      // (returnless: no wait, one writer per slot)
        }
        return false;
    }
    // A one-sided BSDF seen from below evaluates to 0 for every light sample (no shadow ray, nothing added) and its
    // sample() returns 0 (the path ends): nothing after this point can be observed. Only the two-sided
    // transmissive models go on; a normal map can turn a below-the-horizon wi into an above-the-horizon one.
    const float wz = dot(-rd, its.sh.n);                                       // toLocal(its.sh, -rd).z
    bool twoSided = false;
    if (EXT) {
        // modelKey: which code pass B runs for this hit - the model of the row (of the row a normalmap wraps, + 8): pass B deals its lanes by it (kz_wf_shade)
        const int bt = T.bsdfs[its.bsdf].type;
        twoSided = bt == KZ_BSDF_DIELECTRIC || bt == KZ_BSDF_ROUGHDIELECTRIC || bt == KZ_BSDF_NORMALMAP;
        modelKey = (uint32_t)bt;
        if ((EXT & KZ_X_NMAP) && bt == KZ_BSDF_NORMALMAP) modelKey = 8u + ((uint32_t)T.bsdfs[T.bsdfs[its.bsdf].nested].type & 7u);
    }
    survivor = (wz > 0.f) || twoSided || isnan(wz);
    if (survivor && iter >= 3) {
        // Russian roulette (integrator.cpp:237-244) here, in front of the compaction: a path it ends does not take a lane of
        // pass B (at depth 3 and 4 that was 2 of 3 lanes). The survivor's scaled throughput and advanced sampler go back to
        // the path state, where pass B reads them.
        const uint32_t pl = slot / S; const uint32_t pxy = pixList[pl];
        Sampler smp; wfLoadSampler(P, W, slot, (int)(pxy & 0xffffu), (int)(pxy >> 16), sampleBegin + (slot - pl * S), smp, wfPmjDim(P, iter));
        const float4 th = kzLoadStream(&W.thr[slot]);
        V3 throughput = mk(th.x, th.y, th.z);
        const float etaA = compact ? 1.f : th.w;
        const float probability = fminf(maxCoeff(throughput) * etaA * etaA, 0.95f);
        if (probability <= smp.next1D(P, T)) survivor = false;
        else {
            throughput = throughput / probability;
            kzStoreStream(&W.thr[slot], make_float4(throughput.x, throughput.y, throughput.z, th.w));
            wfStoreSampler(P, W, slot, smp);
        }
    } else if (STATS && !survivor && P.nLights > 0) {
        // counters only: the reference draws the roulette sample before it takes the light sample (integrator.cpp:237-247)
        bool alive = true;
        if (iter >= 3) {
            const uint32_t pl = slot / S; const uint32_t pxy = pixList[pl];
            Sampler smp; wfLoadSampler(P, W, slot, (int)(pxy & 0xffffu), (int)(pxy >> 16), sampleBegin + (slot - pl * S), smp, wfPmjDim(P, iter));
            const float4 th = kzLoadStream(&W.thr[slot]);
            const float etaA = compact ? 1.f : th.w;
            if (fminf(maxCoeff(mk(th.x, th.y, th.z)) * etaA * etaA, 0.95f) <= smp.next1D(P, T)) alive = false;
        }
        if (alive) cn.lsamples++;
    }
    return survivor;
}

// -DKZ_SHADE_LATE_POST=1 (the variant DESIGN 9.1 named): pass A classifies on the INTERPOLATED NORMAL alone - the three vertex
// normals and the flag word of the shading record (64 B of its 112) instead of the whole post-intersection - the survivors travel through LDS as the
// 20-B hit record (slot, t, u, v, triangle), and the whole post-intersection (terminator offset, frames, uv) runs in pass B on full waves, where the exact
// back-face test of wfClassify is repeated. Conservative: pass A ends a path only when dot(-d, sum b_i n_i) is negative beyond any rounding of the
// normalisation that the exact test applies afterwards (the sign of a dot product with n / |n| is the sign with n, up to a few ulps of the sum of the
// terms' magnitudes); an emitter hit does the full post-intersection at once (it needs the hit point and the normal, and ends the path).
#ifndef KZ_SHADE_LATE_POST
#define KZ_SHADE_LATE_POST 0
#endif
template <int EXT>
__device__ __forceinline__ bool wfClassifyLight(const KzParams &P, const KzDevTables &T, const KzWf &W, const uint32_t *__restrict__ pixList, uint32_t S, uint32_t sampleBegin,
                                                int iter, bool compact, uint32_t slot, float4 &hitOut, KzSst &sst) {
    const float4 h = kzLoadStream(&W.hit[slot]);
    const float4 rb = kzLoadStream(&W.rayB[slot]);
    const V3 rd = mk(rb.x, rb.y, rb.z);
    hitOut = h;
    if (!(h.x < KZ_INF)) {
        if (iter > 0 && P.bgPresent) {
            const float4 th = kzLoadStream(&W.thr[slot]);
            const V3 c = mk(th.x, th.y, th.z) * backgroundRadiance(P, T, rd);
            unsafeAtomicAdd(W.outR + slot, c.x); unsafeAtomicAdd(W.outG + slot, c.y); unsafeAtomicAdd(W.outB + slot, c.z);
        }
        return false;
    }
    const float4 *sp = reinterpret_cast<const float4 *>(T.shade + __float_as_uint(h.w));
    const float4 s6 = sp[6];
    const uint32_t lf = __float_as_uint(s6.w);
    if ((int32_t)(lf >> 2) - 1 >= 0) {                                            // an emitter: integrator.cpp:226-231, 322-327, exactly as wfClassify does it
        RawHit rh; rh.t = h.x; rh.u = h.y; rh.v = h.z; rh.tri = 0; rh.gid = __float_as_uint(h.w);
        Its its; postIntersect<false>(T, rh, its);
        const KzLightRow &lr = T.lights[its.light];
        const float4 ra = kzLoadStream(&W.rayA[slot]);
        const float4 th = iter == 0 ? make_float4(1.f, 1.f, 1.f, 1.f) : kzLoadStream(&W.thr[slot]);
        const float4 mi = iter == 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : (compact ? make_float4(th.w, 0.f, 0.f, 0.f) : kzLoadStream(&W.misc[slot]));
        const V3 ro = mk(ra.x, ra.y, ra.z);
        const V3 wi = normalized(its.p - ro);
        float bsdfWeight = 1.f;
        if (iter > 0 && mi.z == 0.f) bsdfWeight = powerHeuristic(mi.x, lightPdfSolidAngle(lr.normalization, its.sh.n, wi, its.p, ro));
        if (dot(its.sh.n, -wi) > 0.f) {
            const V3 c = (bsdfWeight * mk(th.x, th.y, th.z)) * mk(lr.radiance[0], lr.radiance[1], lr.radiance[2]);
            unsafeAtomicAdd(W.outR + slot, c.x); unsafeAtomicAdd(W.outG + slot, c.y); unsafeAtomicAdd(W.outB + slot, c.z);
        }
        return false;
    }
    bool twoSided = false;
    if (EXT) { const int bt = T.bsdfs[__float_as_uint(s6.z)].type; twoSided = bt == KZ_BSDF_DIELECTRIC || bt == KZ_BSDF_ROUGHDIELECTRIC || bt == KZ_BSDF_NORMALMAP; }
    bool survivor = true;
    if (!twoSided) {
        V3 n;
        if (lf & 1u) {                                                            // vertex normals: the direction of the shading normal is sum b_i n_i (accel.cpp:177-229)
            const float4 s2 = sp[2], s3 = sp[3], s4 = sp[4];
            const float bx = 1 - (h.y + h.z);
            n = bx * mk(s2.y, s2.z, s2.w) + h.y * mk(s3.x, s3.y, s3.z) + h.z * mk(s3.w, s4.x, s4.y);
        } else {                                                                  // none: the geometric frame (accel.cpp:231-233)
            const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2];
            const V3 p0 = mk(s0.x, s0.y, s0.z);
            n = cross(mk(s0.w, s1.x, s1.y) - p0, mk(s1.z, s1.w, s2.x) - p0);
        }
        const float wzA = -(rd.x * n.x + rd.y * n.y + rd.z * n.z), mag = fabsf(rd.x * n.x) + fabsf(rd.y * n.y) + fabsf(rd.z * n.z);
        if (wzA < -1e-5f * mag) survivor = false;                                 // certainly below the horizon (NaN compares false: survives, as in wfClassify)
    }
    if (survivor && iter >= 3) {                                                  // Russian roulette in front of the compaction, as in wfClassify
        const uint32_t pl = slot / S; const uint32_t pxy = pixList[pl];
        Sampler smp; wfLoadSampler(P, W, slot, (int)(pxy & 0xffffu), (int)(pxy >> 16), sampleBegin + (slot - pl * S), smp, wfPmjDim(P, iter));
        const float4 th = kzLoadStream(&W.thr[slot]);
        V3 throughput = mk(th.x, th.y, th.z);
        const float etaA = compact ? 1.f : th.w;
        const float probability = fminf(maxCoeff(throughput) * etaA * etaA, 0.95f);
        if (probability <= smp.next1D(P, T)) survivor = false;
        else {
            throughput = throughput / probability;
            kzStoreStream(&W.thr[slot], make_float4(throughput.x, throughput.y, throughput.z, th.w));
            wfStoreSampler(P, W, slot, smp);
        }
    }
    sst.mark(0);
    return survivor;
}

// Pass B of shade(iter) for one surviving path: light sample + BSDF eval / pdf + MIS weight -> pending radiance and shadow ray
// (integrator.cpp:247-295), BSDF sample -> throughput and next ray (:297-313). (The roulette of integrator.cpp:237-244 was played in pass A.)
template <bool STATS, int EXT>
__device__ __forceinline__ void wfShadeSurvivor(const KzParams &P, const KzDevTables &T, const KzWf &W, const uint32_t *__restrict__ pixList, uint32_t S, uint32_t sampleBegin,
                                                int iter, bool compact, uint32_t slot, const Its &its, bool &pushNext, bool &pushShadow, Counters &cn, KzSst &sst) {
    const float eps = P.traceBias;
    const float4 rb = kzLoadStream(&W.rayB[slot]);
    const V3 rd = mk(rb.x, rb.y, rb.z);
    float4 th = iter == 0 ? make_float4(1.f, 1.f, 1.f, 1.f) : kzLoadStream(&W.thr[slot]);              // a camera path: generate stores neither (initial values)
    V3 throughput = mk(th.x, th.y, th.z);
    const float eta = compact ? 1.f : th.w;
    float accRough = (iter == 0 || compact) ? 0.f : W.misc[slot].y;
    const uint32_t pl = slot / S;
    const uint32_t pxy = pixList[pl];
    Sampler smp; wfLoadSampler(P, W, slot, (int)(pxy & 0xffffu), (int)(pxy >> 16), sampleBegin + (slot - pl * S), smp, wfPmjDim(P, iter) + (iter >= 3 ? 1u : 0u));
    KzBSDF bsdf = T.bsdfs[its.bsdf];
    NMap nm; surfaceSetup<EXT>(T, its, bsdf, nm);
    const V3 wiLocal = toLocal(its.sh, -rd);
    sst.mark(2);                                    // pass B: record, state loads, sampler setup, BSDF row, frame
    const float pick = smp.next1D(P, T);                                  // drawn even without lights
    if (P.nLights > 0) {                                                  // integrator.cpp:247-295
        const uint32_t li = min((uint32_t)floorf((float)P.nLights * pick), P.nLights - 1);
        const KzLightRow lrow = T.lights[li];
        if (STATS) cn.lsamples++;
        const LightSample ls = lightSample(T, lrow, its.p, [&]() { return smp.next1D(P, T); });
        const V3 lwi = ls.wi; const float dist = ls.dist, lpdf = ls.pdf;
        V3 Ls = ls.Ls;
        Ls = lightPickDivide(P, Ls);
        sst.mark(3);                                // pick + 3 draws + light sample
        const V3 woL = toLocal(its.sh, lwi);
        V3 f; float bpdfL;
        surfEvalPdf<EXT>(bsdf, nm, its, wiLocal, woL, accRough, f, bpdfL);
        const V3 pend = throughput * Ls * f * powerHeuristic(lpdf, bpdfL);
        // a contribution of exactly zero cannot change L whether or not the ray is blocked: skip the ray
        if (!(pend.x == 0.f && pend.y == 0.f && pend.z == 0.f)) {
            kzStoreStream(&W.shA[slot], make_float4(its.p.x, its.p.y, its.p.z, dist - eps));
            kzStoreStream(&W.shB[slot], make_float4(lwi.x, lwi.y, lwi.z, eps));
            kzStoreStream(&W.shL[slot], make_float4(pend.x, pend.y, pend.z, 0.f));
            pushShadow = true;
        }
    }
    sst.mark(4);                                    // eval + pdf towards the light, shadow ray stores
    if (P.regularization && bsdf.type == KZ_BSDF_KAZENSTANDARD) accRough += bsdf.roughness * P.accumulatedRoughness;
    float s2x, s2y; smp.next2D(P, T, s2x, s2y);                           // H1: 2-D before 1-D
    const float s1 = smp.next1D(P, T);
    sst.mark(5);                                    // 2-D + 1-D draw
    V3 woLocal; bool ok, discrete, solid; float etaScale, pdfS;
    const V3 weight = surfSample<EXT>(bsdf, nm, its, wiLocal, accRough, s1, s2x, s2y, woLocal, ok, discrete, etaScale, pdfS, solid);
    sst.mark(6);                                    // BSDF sample
    throughput = throughput * weight;
    const float etaNext = eta * etaScale;
    if (ok && !(weight.x == 0.f && weight.y == 0.f && weight.z == 0.f)) {
        const float bpdf = pdfS >= 0.f ? pdfS : surfPdf<EXT>(bsdf, nm, its, wiLocal, woLocal, accRough, solid);
        const V3 nd = toWorld(its.sh, woLocal);                           // H9
        // the ray after the LAST bounce only matters for the background term
        if (iter + 1 < P.maxDepth || P.bgPresent) {
            kzStoreStream(&W.rayA[slot], make_float4(its.p.x, its.p.y, its.p.z, eps));
            kzStoreStream(&W.rayB[slot], make_float4(nd.x, nd.y, nd.z, KZ_INF));
            kzStoreStream(&W.thr[slot], make_float4(throughput.x, throughput.y, throughput.z, compact ? bpdf : etaNext));
            if (!compact) kzStoreStream(&W.misc[slot], make_float4(bpdf, accRough, discrete ? 1.f : 0.f, 0.f));
            wfStoreSampler(P, W, slot, smp);
            pushNext = true;
        }
    }
}

// The two output queues of a shade kernel, staged in LDS: flush the one(s) that another 256 entries might not fit into (or, at the end,
// whatever is staged). Called by every thread of the workgroup after a barrier that made the staged entries visible.
struct WfQueuePair {
    uint32_t *s_bufN, *s_bufS, *s_nN, *s_nS, *s_gbN, *s_gbS;
    uint32_t *nextQueue, *nextCount, *shadowQueue, *shadowCount;
    uint32_t cap = KZ_WF_QCAP;                              // entries each staging buffer holds
    __device__ __forceinline__ void flush(bool force) {
        const uint32_t nN = *s_nN, nS = *s_nS;
        const bool fN = force ? (nN > 0) : (nN > cap - KZ_BLOCK), fS = force ? (nS > 0) : (nS > cap - KZ_BLOCK);
        if (fN || fS) {                                     // uniform over the workgroup
            if (threadIdx.x == 0) { if (fN) *s_gbN = atomicAdd(nextCount, nN); if (fS) *s_gbS = atomicAdd(shadowCount, nS); }
            __syncthreads();
            if (fN) { const uint32_t gb = *s_gbN; for (uint32_t i = threadIdx.x; i < nN; i += KZ_BLOCK) nextQueue[gb + i] = s_bufN[i]; }
            if (fS) { const uint32_t gb = *s_gbS; for (uint32_t i = threadIdx.x; i < nS; i += KZ_BLOCK) shadowQueue[gb + i] = s_bufS[i]; }
            __syncthreads();
            if (threadIdx.x == 0) { if (fN) *s_nN = 0; if (fS) *s_nS = 0; }      // (kz_wf_shade: the next entries are staged behind the next round's first barrier)
        }
    }
    // Flush both whatever they hold, with the same barriers on every path: for callers that stage the next entries without a barrier of their own
    // (nothing here branches on the shared counts, which a wave that has run ahead may already be adding to again after the last barrier).
    __device__ __forceinline__ void flushAll() {
        const uint32_t nN = *s_nN, nS = *s_nS;
        if (threadIdx.x == 0) { if (nN) *s_gbN = atomicAdd(nextCount, nN); if (nS) *s_gbS = atomicAdd(shadowCount, nS); }
        __syncthreads();
        { const uint32_t gb = *s_gbN; for (uint32_t i = threadIdx.x; i < nN; i += KZ_BLOCK) nextQueue[gb + i] = s_bufN[i]; }
        { const uint32_t gb = *s_gbS; for (uint32_t i = threadIdx.x; i < nS; i += KZ_BLOCK) shadowQueue[gb + i] = s_bufS[i]; }
        __syncthreads();
        if (threadIdx.x == 0) { *s_nN = 0; *s_nS = 0; }
        __syncthreads();
    }
};

// ONE kernel per bounce (KzTune.shadeSplit == 0). Two passes per round of 256 queue entries, because on many scenes about half of the hits end
// the path before any shading happens (a one-sided BSDF seen from behind, an emitter, a miss): pass A rebuilds the intersection record and classifies;
// the entries that still need the light sample and the BSDF sample are compacted through LDS (record = slot + frame + uv),
// and pass B — which carries ~85 % of the kernel's instructions — only ever runs on full waves.
#define KZ_SV_CAP (2 * KZ_BLOCK)
#ifndef KZ_MODEL_SORT
#define KZ_MODEL_SORT 1            // (0: pass B of the EXT variants takes its records in arrival order - the A/B of profiles/r06c_ext)
#endif
// -DKZ_SHADE_CONST_ARGS (development build): the three argument structs of the shade kernel (KzParams 300 B, KzDevTables, KzWf) in
// __constant__ memory, one slot per pass context, instead of in the kernel-argument segment - measured in profiles/r05c_shade_args.
#ifdef KZ_SHADE_CONST_ARGS
struct KzShadeArgs { KzParams P; KzDevTables T; KzWf W; };
__constant__ KzShadeArgs g_kzShadeArgs[KZ_MAX_PASSES_IN_FLIGHT];
#define KZ_SHADE_PARAMS int argSlot
#define KZ_SHADE_BIND const KzParams &P = g_kzShadeArgs[argSlot].P; const KzDevTables &T = g_kzShadeArgs[argSlot].T; const KzWf &W = g_kzShadeArgs[argSlot].W;
#else
#define KZ_SHADE_PARAMS KzParams P, KzDevTables T, KzWf W
#define KZ_SHADE_BIND
#endif
template <bool STATS, int EXT>
__global__ __launch_bounds__(KZ_BLOCK, ((EXT & (KZ_X_TEX | KZ_X_NMAP)) ? 3 : KZ_SHADE_WAVES)) void kz_wf_shade(KZ_SHADE_PARAMS, const uint32_t *__restrict__ pixList, uint32_t S,
                                                        uint32_t sampleBegin, int iter, const uint32_t *__restrict__ queue,
                                                        const uint32_t *__restrict__ countPtr, uint32_t countImm,
                                                        uint32_t *__restrict__ nextQueue, uint32_t *__restrict__ nextCount,
                                                        uint32_t *__restrict__ shadowQueue, uint32_t *__restrict__ shadowCount) {
    KZ_SHADE_BIND
    constexpr bool LATE = KZ_SHADE_LATE_POST && !STATS;      // (the counting variant keeps the exact classification in pass A: its counters follow the reference's order of events)
    constexpr bool NMAPX = (EXT & KZ_X_NMAP) != 0;
    constexpr int SVW = LATE ? 5 : (NMAPX ? 19 : 16);        // words per survivor: slot, p, s, t, n, uv, bsdf row | model key << 24 (+ dpdu: normal maps) | LATE: slot + the hit record
    constexpr uint32_t QCAP = (uint32_t)KZ_WF_QCAP;
    __shared__ uint32_t s_bufN[QCAP], s_bufS[QCAP]; __shared__ uint32_t s_nN, s_nS, s_gbN, s_gbS;
    // The full variant (textures, normal maps) deals the 256 records of pass B to the lanes BY MODEL (a counting sort over 16 keys through LDS: a normal map sorts
    // by the model it wraps, + 8): worth 2.5 % of the kernel on the textured scene, nothing on the scene of eleven constant-parameter models - the items of a pass
    // are pixel-major and a queue keeps that order, so a batch already holds mostly ONE object's hits (profiles/r04q_ext_sorted, r06c_ext/ab_sort_vs_nosort.txt)
    constexpr bool SORT = KZ_MODEL_SORT && NMAPX && !LATE;
    __shared__ uint32_t s_hist[SORT ? 16 : 1]; __shared__ uint16_t s_perm[SORT ? KZ_BLOCK : 1];
    // The survivor table is a stack; its fill count is double-buffered by round (s_svCnt[round & 1]) so that the count for the NEXT round can
    // be written while this round's is still being read: a round then needs two workgroup barriers (records written | records read, output
    // entries staged) instead of six; the two output queues are flushed together, with their own barriers, only when one of them is nearly full.
    __shared__ uint32_t s_sv[SVW * KZ_SV_CAP]; __shared__ uint32_t s_svCnt[2];
    if (threadIdx.x == 0) { s_nN = 0; s_nS = 0; s_svCnt[0] = 0; s_svCnt[1] = 0; }
    if (SORT && threadIdx.x < 16) s_hist[threadIdx.x] = 0;
    __syncthreads();
    WfAppender apN = {s_bufN, &s_nN, &s_gbN, nextQueue, nextCount}, apS = {s_bufS, &s_nS, &s_gbS, shadowQueue, shadowCount};
    WfQueuePair qp = {s_bufN, s_bufS, &s_nN, &s_nS, &s_gbN, &s_gbS, nextQueue, nextCount, shadowQueue, shadowCount, QCAP};
    const uint32_t count = countPtr ? *countPtr : countImm;
    const int lane = threadIdx.x & 63;
    // Path state between bounces. The lean variant has no BSDF that changes eta or samples a discrete lobe, and without regularisation the
    // accumulated roughness stays 0: the only thing `misc` would carry is the pdf of the sampled direction, which then rides in the free fourth
    // word of the throughput record (16 B less to write and to read back per bounce and path; the state arrays are the kernel's HBM traffic).
    const bool compact = !EXT && !P.regularization;
    Counters cn = {0, 0, 0, 0, 0, 0};
    KzSst sst;
    uint32_t round = 0;
    for (uint32_t base = blockIdx.x * KZ_BLOCK;; base += gridDim.x * KZ_BLOCK, ++round) {
        const bool more = base < count;                                               // uniform over the workgroup
        sst.mark(9);
        // ================= pass A: hit record -> intersection, miss / emitter / back-face end here =================
        bool survivor = false;
        uint32_t slot = 0;
        Its its;
        uint32_t modelKey = 0;
        float4 hrec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (more && base + threadIdx.x < count) {
            const uint32_t qi = base + threadIdx.x;
            slot = queue ? queue[qi] : qi;
            if (LATE) survivor = wfClassifyLight<EXT>(P, T, W, pixList, S, sampleBegin, iter, compact, slot, hrec, sst);
            else survivor = wfClassify<STATS, EXT>(P, T, W, pixList, S, sampleBegin, iter, compact, slot, its, modelKey, cn, sst);
        }
        sst.mark(0);                                        // pass A
        {   // compaction of the survivors onto the LDS record stack
            const unsigned long long m = __ballot(survivor);
            uint32_t b = 0;
            if (lane == 0 && m) b = atomicAdd(&s_svCnt[round & 1u], (uint32_t)__popcll(m));
            b = __shfl(b, 0, 64);
            if (survivor) {
                const uint32_t e = b + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                uint32_t *r = s_sv + e;
                if (LATE) { r[0] = slot; r[KZ_SV_CAP] = __float_as_uint(hrec.x); r[2 * KZ_SV_CAP] = __float_as_uint(hrec.y); r[3 * KZ_SV_CAP] = __float_as_uint(hrec.z); r[4 * KZ_SV_CAP] = __float_as_uint(hrec.w); }
                else {
                r[0] = slot; r[KZ_SV_CAP] = __float_as_uint(its.p.x); r[2 * KZ_SV_CAP] = __float_as_uint(its.p.y); r[3 * KZ_SV_CAP] = __float_as_uint(its.p.z);
                r[4 * KZ_SV_CAP] = __float_as_uint(its.sh.s.x); r[5 * KZ_SV_CAP] = __float_as_uint(its.sh.s.y); r[6 * KZ_SV_CAP] = __float_as_uint(its.sh.s.z);
                r[7 * KZ_SV_CAP] = __float_as_uint(its.sh.t.x); r[8 * KZ_SV_CAP] = __float_as_uint(its.sh.t.y); r[9 * KZ_SV_CAP] = __float_as_uint(its.sh.t.z);
                r[10 * KZ_SV_CAP] = __float_as_uint(its.sh.n.x); r[11 * KZ_SV_CAP] = __float_as_uint(its.sh.n.y); r[12 * KZ_SV_CAP] = __float_as_uint(its.sh.n.z);
                r[13 * KZ_SV_CAP] = __float_as_uint(its.uvx); r[14 * KZ_SV_CAP] = __float_as_uint(its.uvy); r[15 * KZ_SV_CAP] = EXT ? (its.bsdf | (modelKey << 24)) : its.bsdf;      // (kz_scene_create: < 2^24 rows)
                if (NMAPX) { r[16 * KZ_SV_CAP] = __float_as_uint(its.dpdu.x); r[17 * KZ_SV_CAP] = __float_as_uint(its.dpdu.y); r[18 * KZ_SV_CAP] = __float_as_uint(its.dpdu.z); }
                }
            }
        }
        __syncthreads();
        sst.mark(1);                                        // compaction + barrier
        // ================= pass B: one survivor per thread once a full workgroup of them is waiting (or at the end) =================
        // Before pass A at most 255 records wait, pass A adds at most 256: the table (512) cannot overflow.
        const uint32_t n0 = s_svCnt[round & 1u];
        bool pushNext = false, pushShadow = false;
        const uint32_t take = n0 >= KZ_BLOCK ? (uint32_t)KZ_BLOCK : (more ? 0u : n0);
        if (threadIdx.x == 0) s_svCnt[(round + 1u) & 1u] = n0 - take;        // what the next round's pass A appends to (visible behind the barrier below)
        uint32_t mine = threadIdx.x;                                          // which of the `take` records this thread shades
        if (SORT && take > 64u) {                                             // (uniform; one wave's worth needs no dealing)
            // counting sort of the taken records by model key: rank within the key by an LDS atomic, bucket bases by a 16-entry prefix, scatter of the record numbers
            uint32_t key = 0, rank = 0;
            if (threadIdx.x < take) { key = s_sv[15 * KZ_SV_CAP + (n0 - take) + threadIdx.x] >> 24; rank = atomicAdd(&s_hist[key & 15u], 1u); }
            __syncthreads();
            if (threadIdx.x < take) {
                uint32_t basePos = 0;
                for (uint32_t k = 0; k < (key & 15u); ++k) basePos += s_hist[k];
                s_perm[basePos + rank] = (uint16_t)threadIdx.x;
            }
            __syncthreads();
            if (threadIdx.x < take) mine = s_perm[threadIdx.x];
        }
        if (threadIdx.x < take) {
            const uint32_t *r = s_sv + (n0 - take) + mine;
            slot = r[0];
            bool alive = true;
            if (LATE) {
                // the whole post-intersection on a full wave, then wfClassify's exact test (pass A kept every hit it could not rule out)
                RawHit rh; rh.t = __uint_as_float(r[KZ_SV_CAP]); rh.u = __uint_as_float(r[2 * KZ_SV_CAP]); rh.v = __uint_as_float(r[3 * KZ_SV_CAP]); rh.tri = 0; rh.gid = r[4 * KZ_SV_CAP];
                postIntersect<false>(T, rh, its);
                const float4 rb = kzLoadStream(&W.rayB[slot]);
                const float wz = dot(-mk(rb.x, rb.y, rb.z), its.sh.n);
                bool twoSided = false;
                if (EXT) { const int bt = T.bsdfs[its.bsdf].type; twoSided = bt == KZ_BSDF_DIELECTRIC || bt == KZ_BSDF_ROUGHDIELECTRIC || bt == KZ_BSDF_NORMALMAP; }
                alive = (wz > 0.f) || twoSided || isnan(wz);
            } else {
            its.p = mk(__uint_as_float(r[KZ_SV_CAP]), __uint_as_float(r[2 * KZ_SV_CAP]), __uint_as_float(r[3 * KZ_SV_CAP]));
            its.sh.s = mk(__uint_as_float(r[4 * KZ_SV_CAP]), __uint_as_float(r[5 * KZ_SV_CAP]), __uint_as_float(r[6 * KZ_SV_CAP]));
            its.sh.t = mk(__uint_as_float(r[7 * KZ_SV_CAP]), __uint_as_float(r[8 * KZ_SV_CAP]), __uint_as_float(r[9 * KZ_SV_CAP]));
            its.sh.n = mk(__uint_as_float(r[10 * KZ_SV_CAP]), __uint_as_float(r[11 * KZ_SV_CAP]), __uint_as_float(r[12 * KZ_SV_CAP]));
            its.uvx = __uint_as_float(r[13 * KZ_SV_CAP]); its.uvy = __uint_as_float(r[14 * KZ_SV_CAP]); its.bsdf = EXT ? (r[15 * KZ_SV_CAP] & 0xFFFFFFu) : r[15 * KZ_SV_CAP];
            if (NMAPX) its.dpdu = mk(__uint_as_float(r[16 * KZ_SV_CAP]), __uint_as_float(r[17 * KZ_SV_CAP]), __uint_as_float(r[18 * KZ_SV_CAP]));
            }
            if (alive) wfShadeSurvivor<STATS, EXT>(P, T, W, pixList, S, sampleBegin, iter, compact, slot, its, pushNext, pushShadow, cn, sst);
        }
        sst.mark(7);                                        // next-ray stores (and, for lanes without a survivor, nothing)
        apN.push(pushNext, slot); apS.push(pushShadow, slot);
        __syncthreads();                                   // every record of this batch has been read, the output entries are staged
        if (SORT && threadIdx.x < 16) s_hist[threadIdx.x] = 0;     // (read for the last time in front of the barrier above; counted into again behind the next round's first barrier)
        qp.flush(false);
        sst.mark(8);                                        // barrier + queue staging + flushes
        if (!more && n0 - take == 0) break;
        if (!more) base -= gridDim.x * KZ_BLOCK;           // stay past the end while the table drains
    }
    __syncthreads();
    qp.flush(true);
    if (STATS) wfStatsFlush(W.stats, cn, 0);
    sst.flush(W.stats);
}



// ---- persistent traversal kernel -----------------------------------------------------------------------------------------
// One wave keeps 64 rays in flight. Lanes whose ray has finished are refilled from a wave-local pool of queue entries
// (one global atomic per tune.batch rays) once fewer than tune.refill lanes are busy, so a long ray no longer
// idles the other 63 lanes until the end of its chunk. The loop is "while-while": all lanes descend inner nodes together,
// then all lanes at a leaf test triangles together; a few stragglers still descending do not hold up the leaf phase.
// The tree is the quantised BVH4 (KzNode4). The per-lane stack keeps tune.ldsStack entries in LDS ([entry][lane] columns,
// conflict free) and spills deeper entries to a global overflow area sized from the builder's worst-case bound, so LDS
// does not cap occupancy.
// MODE 0: closest hit -> W.hit; 1: same on the shA/shB ray, previous hit kept on a miss (H6); 2: shadow test -> adds W.shL,
// with the reference's closest-hit walk-through (integrator.cpp:257-278) for the lanes that need it ("literal" lanes);
// 4: MODE 2 without the walk-through machinery — a shadow ray whose segment crosses a triangle of an invisible light (rare) is
// not traced here but appended to the queue passed in queueB / countPtrB, which a MODE 2 launch takes afterwards. With `literal`
// constant false the closest-hit bookkeeping of the shadow lanes (hit distance, barycentrics, triangle ids) disappears from that
// instantiation: 57 VGPRs and no spills instead of 64 with 9 spilled.
// Shadow test (exact, see kz_devfn.h shadowOccluded): any-hit unless an invisible-light triangle lies on the segment.
// (The BVH2 form, the per-lane key stack, the LDS top-of-tree and the mixed launches of round 2 live in kz_experiments.h.)
#ifndef KZ_TRACE_WAVES
#define KZ_TRACE_WAVES 8            // waves per SIMD the per-lane traversal is compiled for (64 VGPRs); 7 = 72 VGPRs measured in r02i
#endif
template <int MODE, bool STATS>
__global__ __launch_bounds__(KZ_BLOCK) __attribute__((amdgpu_waves_per_eu(STATS ? KZ_TRACE_WAVES - 1 : KZ_TRACE_WAVES, STATS ? KZ_TRACE_WAVES - 1 : KZ_TRACE_WAVES)))
void kz_wf_trace(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ queue,
                                                        const uint32_t *__restrict__ countPtr, uint32_t countImm, uint32_t *__restrict__ head, KzTune tune,
                                                        uint32_t *__restrict__ queueB, uint32_t *__restrict__ countPtrB) {
    extern __shared__ uint32_t s_stack[];
    const uint32_t count = countPtr ? *countPtr : countImm;
    constexpr bool SHADOW = MODE == 2 || MODE == 4;
    constexpr int kind = MODE == 4 ? 2 : MODE;                        // ray kind: 0 closest hit, 1 closest hit on the walk-through ray, 2 shadow
    const int lane = threadIdx.x & 63;
    // The lane's stack is a column of the [entry][lane] LDS array (+ one scratch row). Its state is `top`, the LDS BYTE ADDRESS of the next
    // free entry: pushes and pops are ds_write / ds_read at that address, one v_add_u32 apart (an entry index costs a v_lshl_add_u32 - a
    // 4-cycle instruction on gfx950, scripts/micro/valu_ops.hip - per access).
    constexpr uint32_t rowB = KZ_BLOCK * 4u;
    const uint32_t stkBase = kzLdsAddr(s_stack + threadIdx.x), ldsRows = (uint32_t)tune.ldsStack * rowB;
    uint32_t top = stkBase;
    // overflow rows live at tune.ovf[row * ovfStride + thread]: the (rare) deep case forms its address from the scalar base and a 32-bit
    // thread index instead of keeping a 64-bit per-lane pointer alive through the loop
    uint32_t *const ovfBase = tune.ovf;
    const uint32_t ovfLane = blockIdx.x * KZ_BLOCK + threadIdx.x;
    const size_t ovfStride = tune.ovfStride;
#define ovf(row) ovfBase[(size_t)(row) * ovfStride + ovfLane]
    auto push = [&](uint32_t v) {
        const uint32_t dB = top - stkBase;
        if (dB < ldsRows) kzLdsPut(top, v); else ovf((dB - ldsRows) / rowB) = v;
        top += rowB;
    };
    const uint32_t root = P.rootRef4;
    const float eps = P.traceBias;
    Counters cn = {0, 0, 0, 0, 0, 0};
    // Queue entries are claimed in batches. The FIRST batch of a wave is static (wave w owns entries [w*batch, (w+1)*batch)): a
    // launch on a short queue then costs no atomics at all, where 8192 waves hitting one counter took ~95 us (one word serves ~88
    // dequeues/us) - the whole duration of the late, nearly empty bounces of a small frame. Further batches come from the shared counter.
    const uint32_t nWaves = gridDim.x * (KZ_BLOCK / 64), waveId = blockIdx.x * (KZ_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t batch = max(64u, min((uint32_t)tune.batch, ((count + nWaves - 1) / nWaves + 63u) & ~63u));   // short queues: spread over all waves
    const bool staticOnly = (unsigned long long)nWaves * batch >= count;
    uint32_t poolNext = min(waveId * batch, count), poolEnd = min(poolNext + batch, count);
    // Guided reservations: a wave asks for a quarter of its even share of what is LEFT (never less than `batch`, never more than 8 of them). The shared
    // counter serves ~88 reservations per microsecond; rays that end quickly (the shadow rays of an open scene) otherwise queue up behind it, and a
    // fixed larger batch pays for it at the tail of a launch, where the last waves sit on the last big batches.
    uint32_t resv = batch;
    bool exhausted = false;
    bool active = false, literal = false;
    V3 o = mk(0.f), d = mk(0.f);
    float rx = 0.f, ry = 0.f, rz = 0.f, tmin = 0.f, tmax = 0.f, segMax = 0.f;
    uint32_t cur = 0, slot = 0;
    bool found = false; float bt = 0.f, bu = 0.f, bv = 0.f; uint32_t btri = 0, bgid = 0;

    // next stack entry -> cur; false when the stack is empty
    auto popNext = [&]() -> bool {
        if (top == stkBase) return false;
        top -= rowB;
        const uint32_t dB = top - stkBase;
        // LDS read of the entry (of the scratch row in the rare deep case, where the global entry replaces it). The address-space-3 form
        // matters: through generic pointers the compiler merged the two loads into ONE flat_load at a selected address - a vector-memory
        // instruction per pop beside the node fetches, where a ds_read costs that pipeline nothing.
        uint32_t v = kzLdsGet(min(top, stkBase + ldsRows));
        if (dB >= ldsRows) v = ovf((dB - ldsRows) / rowB);
        cur = v;
        return true;
    };
#ifdef KZ_LANESTAT
    // development build only (-DKZ_LANESTAT): where the lanes of the while-while loop are, summed per wave (wave-uniform counts)
    unsigned long long lsNodeIters = 0, lsActiveAtNode = 0, lsInnerAtNode = 0, lsLeafPhases = 0, lsLeafLanes = 0, lsRefills = 0, lsRefillLanes = 0, lsTriIters = 0;
#endif
    // An unoccluded shadow ray adds its pending radiance to the path's sums. As plain read-modify-writes that was three dependent HBM round
    // trips in the middle of the loop (load, wait, add, store - per colour, the compiler keeps them apart), with the whole wave waiting each time a
    // lane's ray ended: the shadow kernel ran at VALU busy 0.75 where the closest-hit kernel, which only stores, reaches 0.99. The pending radiance
    // now arrives with the ray at the refill (MODE 4) and the sums take it as returnless float atomics (one writer per slot: the same single
    // rounding as the add): nothing to wait for.
    float plR = 0.f, plG = 0.f, plB = 0.f;
    auto addPending = [&]() {
        uint32_t s_ = slot; asm volatile("" : "+v"(s_));          // (the three addresses are formed here: hoisted to the refill they were six registers held through the loop)
        if (MODE != 4) { const float4 l = W.shL[s_]; plR = l.x; plG = l.y; plB = l.z; }
        unsafeAtomicAdd(W.outR + s_, plR); unsafeAtomicAdd(W.outG + s_, plG); unsafeAtomicAdd(W.outB + s_, plB);
    };
    // the lane's stack ran empty: publish the result (or, for a literal shadow lane, decide / walk through the light)
    auto finish = [&]() {
        active = false;
        if (kind == 0) kzStoreStream(&W.hit[slot], make_float4(bt, bu, bv, __uint_as_float(bgid)));      // (without a hit the record still holds what the refill put there: +inf, 0, 0, 0 = the miss record)
        if (kind == 1) { if (found) kzStoreStream(&W.hit[slot], make_float4(bt, bu, bv, __uint_as_float(bgid))); }
        if (kind == 2) {
            if (MODE == 4 || !literal || !found) addPending();                       // nothing on the segment
            else {
                const uint32_t om = __float_as_uint(reinterpret_cast<const float4 *>(T.tris + btri)[2].y);      // (the leaf triangle is in cache; its shading record is not)
                const int ol = T.meshes[om].light;
                if (ol >= 0 && !T.lights[ol].primaryVisibility) {                     // walk through (integrator.cpp:273-274)
                    o = o + d * (bt + eps); tmin = eps; segMax = segMax - bt; tmax = segMax;
                    found = false; bt = KZ_INF; cur = root; top = stkBase; asm volatile("" : "+v"(top)); active = true;
                    if (STATS) cn.rays++;
                }
            }
        }
    };

#ifdef KZ_TRACESTAT
    // development build only (-DKZ_TRACESTAT): wall-clock cycles (s_memtime, per wave) of the refill / node / leaf parts of the loop
    unsigned long long tsT = __builtin_amdgcn_s_memtime(), tsAcc[4] = {0, 0, 0, 0}, tsTri = 0;
#define KZ_TST(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsAcc[k] += t_ - tsT; tsT = t_; } while (0)
#else
#define KZ_TST(k) do { } while (0)
#endif
    for (;;) {
        KZ_TST(3);
        // ---- refill idle lanes
        const unsigned long long act = __ballot(active);
        const int nAct = __popcll(act);
        if (nAct < tune.refill && !exhausted) {
            if (poolNext >= poolEnd) {
                uint32_t b = count;
                if (!staticOnly) {
                    if (lane == 0) b = atomicAdd(head, resv);
                    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    b = (b < 0xFFFFFFFFu - nWaves * batch) ? b + nWaves * batch : count;            // dynamic batches start behind the static ones
                }
                if (b >= count) { exhausted = true; poolNext = poolEnd = 0; }
                else {
                    poolNext = b; poolEnd = min(b + resv, count);
                    resv = min(8u * batch, max(batch, (((count - poolEnd) / nWaves) >> 2) & ~63u));
                }
            }
            if (!exhausted) {
                const uint32_t take = min((uint32_t)(64 - nAct), poolEnd - poolNext);
#ifdef KZ_LANESTAT
                lsRefills++; lsRefillLanes += take;
#endif
                const unsigned long long idle = ~act;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));   // idle lanes below this one
                if (!active && rank < take) {
                    const uint32_t qi = poolNext + rank;
                    slot = queue ? kzLoadStream(queue + qi) : qi;
                    // (an idle lane's stack is empty, or abandoned by an occluded shadow ray. The empty asm makes the new top a value of its own: as a
                    // plain copy of stkBase the compiler deferred the copy down some of the refill's paths and reused the register on the others -
                    // the instantiations with counters walked garbage stacks, profiles/r03u_valu_ops)
                    top = stkBase; asm volatile("" : "+v"(top));
                    float4 a, b;
                    if (kind == 0) { a = kzLoadStream(&W.rayA[slot]); b = kzLoadStream(&W.rayB[slot]); }
                    else { const float4 sa = kzLoadStream(&W.shA[slot]), sb = kzLoadStream(&W.shB[slot]); a = make_float4(sa.x, sa.y, sa.z, sb.w); b = make_float4(sb.x, sb.y, sb.z, sa.w);
                           if (MODE == 4) { const float4 l = kzLoadStream(&W.shL[slot]); plR = l.x; plG = l.y; plB = l.z; } }
                    o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z); tmin = a.w; tmax = b.w; segMax = b.w;
                    if (MODE == 2) asm volatile("" : "+v"(segMax));          // (a value of its own, not a deferred copy of tmax's register: see `top` above)
                    found = false; bt = KZ_INF; bu = bv = 0.f; btri = 0; bgid = 0; literal = false;
                    if (STATS && MODE != 4) cn.rays++;
                    if ((root != 0xFFFFFFFFu) && rayIsFinite(o, d)) {
                        // The FMA slab form q*(s*rcp) + (p-o)*rcp turns into inf - inf = NaN for a zero direction component,
                        // which would switch that axis off (a huge slab of the tree gets walked). A tiny signed stand-in keeps
                        // every product finite; the sign of (box - origin) * 1e20 still decides the slab exactly as 1/0 would.
                        rx = rcpExact(fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x);
                        ry = rcpExact(fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y);
                        rz = rcpExact(fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z);
                        cur = root; active = true;
                        if (MODE == 4) {
                            if (invisibleLightOnSegment(P, T, o, d, rx, ry, rz, tmin, tmax)) {       // (launched only when P.shadowFast)
                                queueB[atomicAdd(countPtrB, 1u)] = slot;
                                active = false;
                            } else if (STATS) cn.rays++;
                        } else if (kind == 2) literal = !P.shadowFast || invisibleLightOnSegment(P, T, o, d, rx, ry, rz, tmin, tmax);
                    } else {
                        // a ray that cannot hit anything (empty scene, non-finite origin/direction)
                        // (the constants are made HERE: hoisted out of the loop they took four registers for the whole kernel, then a spill slot)
                        if (kind == 0) { float inf = KZ_INF, zero = 0.f; asm volatile("" : "+v"(inf), "+v"(zero)); kzStoreStream(&W.hit[slot], make_float4(inf, zero, zero, zero)); }
                        if (kind == 2) addPending();
                        if (STATS && MODE == 4) cn.rays++;
                    }
                }
                poolNext += take;
            }
        }
        KZ_TST(0);
        if (!__any(active)) { if (exhausted) break; continue; }
        // ---- node phase: descend until (almost) every busy lane holds a leaf
        for (;;) {
            const bool inner = active && !(cur & 0x80000000u);
            const unsigned long long im = __ballot(inner);
            if (im == 0) break;
            if (__popcll(im) < tune.postpone && __ballot(active && (cur & 0x80000000u)) != 0) break;
#ifdef KZ_LANESTAT
            lsNodeIters++; lsInnerAtNode += __popcll(im); lsActiveAtNode += __popcll(__ballot(active));
#endif
            if (inner) {
                if (STATS) cn.nodes++;
                // Child order. Rays descend into the NEAREST hit child and leave the other hit children on the stack in slot order (a full
                // sort of the siblings buys 6 % fewer node visits and costs 12 % more instructions per visit). Any-hit shadow rays took the
                // hit children in slot order altogether until the pushes got cheap (round 3): in dense geometry the nearest child is where
                // the occluder is, and the selection now costs less than the visits it saves. Pushes are branch-free (below).
                uint32_t key[4]; uint4 refs;
                bool p0, p1, p2, p3;                                          // child i goes on the stack
                uint32_t nxt; bool any;
#ifndef KZ_SHADOW_ORDERED
#define KZ_SHADOW_ORDERED 1             // any-hit rays descend into the nearest hit child first too: 0 = the slot order of round 2 (C4 shadow stage 21.2 ms, C3 21.4; 1: 20.3, 21.6)
#endif
                constexpr bool SLOTORDER = SHADOW && !KZ_SHADOW_ORDERED;
                node4Keys<!SLOTORDER>(T, cur, o, rx, ry, rz, tmin, tmax, key, refs);
                if (SLOTORDER) {
                    const bool h0 = key[0] != 0xFFFFFFFFu, h1 = key[1] != 0xFFFFFFFFu, h2 = key[2] != 0xFFFFFFFFu, h3 = key[3] != 0xFFFFFFFFu;
                    any = h0 || h1 || h2 || h3;
                    nxt = h0 ? refs.x : (h1 ? refs.y : (h2 ? refs.z : refs.w));
                    p0 = false; p1 = h1 && h0; p2 = h2 && (h0 || h1); p3 = h3 && (h0 || h1 || h2);
                } else {
                    const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
                    any = kmin != 0xFFFFFFFFu;
                    nxt = pick4b(refs, kmin);
                    p0 = key[0] != 0xFFFFFFFFu && key[0] != kmin; p1 = key[1] != 0xFFFFFFFFu && key[1] != kmin;
                    p2 = key[2] != 0xFFFFFFFFu && key[2] != kmin; p3 = key[3] != 0xFFFFFFFFu && key[3] != kmin;
                }
                if (top - stkBase + 3u * rowB <= ldsRows) {                      // common case: everything stays in LDS
                    // every child is written at the running top, which advances only past the children that stay: one that does not is
                    // overwritten by the next store or left above the top (the last row it can reach is the scratch row). No address selects.
                    uint32_t a = top;
                    if (!SLOTORDER) { kzLdsPut(a, refs.x); a += p0 ? rowB : 0u; }
                    kzLdsPut(a, refs.y); a += p1 ? rowB : 0u;
                    kzLdsPut(a, refs.z); a += p2 ? rowB : 0u;
                    kzLdsPut(a, refs.w); a += p3 ? rowB : 0u;
                    top = a;
                } else {
                    if (p0) push(refs.x);
                    if (p1) push(refs.y);
                    if (p2) push(refs.z);
                    if (p3) push(refs.w);
                }
                if (any) cur = nxt;
                else if (!popNext()) finish();
            }
        }
        KZ_TST(1);
        // ---- leaf phase
#ifdef KZ_LANESTAT
        { const unsigned long long lm = __ballot(active && (cur & 0x80000000u)); if (lm) { lsLeafPhases++; lsLeafLanes += __popcll(lm);
          uint32_t mc = (active && (cur & 0x80000000u)) ? (cur & 7u) + 1 : 0; for (int off = 32; off > 0; off >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, off, 64)); lsTriIters += mc; } }
#endif
        if (active && (cur & 0x80000000u)) {
            const uint32_t start = (cur & 0x7fffffffu) >> 3, cnt = (cur & 7u) + 1;
            bool occluded = false;
            for (uint32_t i = 0; i < cnt; ++i) {
                float t, u, v; uint32_t g;
                if (STATS) cn.tris++;
                if (!triTest(T.tris + start + i, o, d, tmin, tmax, t, u, v, g)) continue;
                if (kind == 2 && (MODE == 4 || !literal)) { occluded = true; break; }   // any hit blocks: nothing to add
                if (!found || t < bt || (t == bt && g < bgid)) { found = true; bt = t; bu = u; bv = v; btri = start + i; bgid = g; tmax = t; }
            }
#ifdef KZ_TRACESTAT
            { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tsTri += t_ - tsT; }
#endif
            if (occluded) active = false;
            else if (!popNext()) finish();
        }
        KZ_TST(2);
    }
    if (STATS) wfStatsFlush(W.stats, cn, 0);
#ifdef KZ_TRACESTAT
    if (lane == 0) { for (int k = 0; k < 4; ++k) atomicAdd(W.stats + 8 + (SHADOW ? 4 : 0) + k, tsAcc[k]); atomicAdd(W.stats + 16 + (SHADOW ? 1 : 0), tsTri); }
#endif
#ifdef KZ_LANESTAT
    if (lane == 0) {
        unsigned long long *ls = W.stats + 8 + (SHADOW ? 8 : 0);
        atomicAdd(ls + 0, lsNodeIters); atomicAdd(ls + 1, lsActiveAtNode); atomicAdd(ls + 2, lsInnerAtNode); atomicAdd(ls + 3, lsLeafPhases);
        atomicAdd(ls + 4, lsLeafLanes); atomicAdd(ls + 5, lsRefills); atomicAdd(ls + 6, lsRefillLanes); atomicAdd(ls + 7, lsTriIters);
    }
#endif
#undef ovf
}

// ---- packet traversal for coherent rays (the camera rays of a pass) ------------------------------------------------------
// The items of a pass are ordered pixel-major (item = pixel * S + sample, pixels in 8x8 blocks), so the 64 camera rays of a wave
// belong to one pixel (S >= 64) or to a few neighbouring ones: they walk the same nodes. The per-lane kernel above spends ~45 % of
// its node step on per-lane child sorting and stack traffic and keeps 43 of 64 lanes busy on these rays
// (profiles/r01j_big_passes); here the WAVE walks the tree once with ONE stack:
//   * `cur` and the stack are wave-uniform (SGPRs; the stack is a VGPR written and read with v_writelane / v_readlane, one entry
//     per lane position: 128 entries in two registers, enough for the builder's worst case or the launch falls back to kz_wf_trace);
//   * every lane tests the node's four boxes against ITS ray and tmax (node4Keys, unchanged); the v_cmp results are the
//     wave's hit masks; a child is visited if ANY lane hit it, in the front-to-back order of the first lane that hit anything;
//   * at a leaf every lane runs the same Moeller-Trumbore test (triTest) and keeps its own closest hit.
// Every lane sees a superset of the nodes its own traversal would visit and the boxes are conservative, so each lane's closest
// hit (t, u, v, triangle; ties to the lower triangle id) is the one kz_wf_trace finds: the films stay bit-identical.

// v_writelane_b32: one lane of a VGPR takes a wave-uniform value (this clang has a builtin for v_readlane only). On gfx9 the
// instruction may read one SGPR besides M0, so the lane select travels in M0 (saved and restored around it).
__device__ __forceinline__ void kzWriteLane(uint32_t &reg, uint32_t value, int laneSel) {
    uint32_t keep;                                   // M0 is reserved by the compiler: put it back
    asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1" : "+v"(reg), "=&s"(keep) : "s"(value), "s"(laneSel));
}

// Every lane also keeps, for the first KZ_PACKET_KEYS entries of the shared stack, ITS OWN entry distance of that box (LDS,
// [entry][lane]); a popped entry that no lane can still reach (box behind the lane's closest hit, or never hit by the lane) is dropped
// for one LDS read and a compare instead of a node step.
// FIX (scenes with a light of lightPrimaryVisibility == false): the epilogue also does what kz_wf_primary_fix does for the per-lane
// path - a first hit on such a light gets its continuation ray (integrator.cpp:214-219) and its slot goes to fixQueue, which a
// kz_wf_trace<1> launch takes next. The hit's triangle id is in a register here; a range test on it (KzParams.ilGidLo / ilGidSpan)
// keeps every other hit away from the shading-record gather, so the separate pass over all hit records (2.1 GB per 2^27-item pass)
// and its launch on the pass's critical path are gone.
#define KZ_PACKET_KEYS 16
template <bool STATS, bool FIX>
__global__ __launch_bounds__(KZ_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void kz_wf_trace_packet(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ queue,
                                                        const uint32_t *__restrict__ countPtr, uint32_t countImm, uint32_t *__restrict__ head, int batchPackets,
                                                        uint32_t *__restrict__ fixQueue, uint32_t *__restrict__ fixCount) {
    constexpr bool KEYS = true;
    __shared__ uint32_t s_keys[KZ_PACKET_KEYS * KZ_BLOCK];
    uint32_t *kst = s_keys + threadIdx.x;
    const uint32_t count = countPtr ? *countPtr : countImm;
    const uint32_t nPackets = (count + 63u) / 64u;
    const int lane = threadIdx.x & 63;
    const uint32_t root = P.rootRef4;
    Counters cn = {0, 0, 0, 0, 0, 0};
    // first batch static (wave w owns packets [w*bp, (w+1)*bp)), further batches from the shared counter: see kz_wf_trace
    const uint32_t nWaves = gridDim.x * (KZ_BLOCK / 64), waveId = blockIdx.x * (KZ_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t bp = max(1u, min((uint32_t)batchPackets, (nPackets + nWaves - 1) / nWaves));
    const bool staticOnly = (unsigned long long)nWaves * bp >= nPackets;
    uint32_t pNext = min(waveId * bp, nPackets), pEnd = min(pNext + bp, nPackets);
    for (;;) {
        if (pNext >= pEnd) {
            if (staticOnly) break;
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(head, bp);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b) + nWaves * bp;
            if (b >= nPackets) break;
            pNext = b; pEnd = min(b + bp, nPackets);
        }
        const uint32_t qi = pNext * 64u + (uint32_t)lane;
        ++pNext;
        const bool have = qi < count;
        const uint32_t slot = have ? (queue ? queue[qi] : qi) : 0u;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 1.f, 0.f);
        if (have) { a = kzLoadStream(&W.rayA[slot]); b = kzLoadStream(&W.rayB[slot]); }
        const V3 o = mk(a.x, a.y, a.z), d = mk(b.x, b.y, b.z);
        const float tmin = a.w;
        // a lane without a ray, or with a ray that can hit nothing (non-finite, see kz_wf_trace), carries tmax = -inf: it fails every box and triangle test
        const bool live = have && root != 0xFFFFFFFFu && rayIsFinite(o, d);
        float tmax = live ? b.w : -KZ_INF;
        const float rx = rcpExact(fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x);
        const float ry = rcpExact(fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y);
        const float rz = rcpExact(fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z);
        bool found = false; float bt = KZ_INF, bu = 0.f, bv = 0.f; uint32_t bgid = 0;
        if (STATS && have) cn.rays++;
        if (__ballot(live) != 0ull) {
            uint32_t cur = root;                       // wave-uniform
            int sp = 0;                                // wave-uniform
            uint32_t stk0 = 0, stk1 = 0;               // the shared stack: entry i lives in lane i of stk0 (i < 64) or lane i - 64 of stk1
            for (;;) {
                if (!(cur & 0x80000000u)) {
                    uint32_t key[4]; uint4 refs;
                    node4Keys(T, cur, o, rx, ry, rz, tmin, tmax, key, refs);
                    if (STATS && live) cn.nodes++;
                    const unsigned long long m0 = __ballot(key[0] != 0xFFFFFFFFu), m1 = __ballot(key[1] != 0xFFFFFFFFu),
                                             m2 = __ballot(key[2] != 0xFFFFFFFFu), m3 = __ballot(key[3] != 0xFFFFFFFFu);
                    const unsigned long long any = m0 | m1 | m2 | m3;
                    bool descended = false;
                    if (any != 0ull) {
                        // order of the first lane that hit anything; a child only other lanes hit sorts behind that lane's own hits
                        const int rep = __ffsll((long long)any) - 1;
                        uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key[0], rep), k1 = (uint32_t)__builtin_amdgcn_readlane((int)key[1], rep),
                                 k2 = (uint32_t)__builtin_amdgcn_readlane((int)key[2], rep), k3 = (uint32_t)__builtin_amdgcn_readlane((int)key[3], rep);
                        k0 = m0 ? min(k0, 0xFFFFFFF8u) : 0xFFFFFFFFu; k1 = m1 ? min(k1, 0xFFFFFFF9u) : 0xFFFFFFFFu;
                        k2 = m2 ? min(k2, 0xFFFFFFFAu) : 0xFFFFFFFFu; k3 = m3 ? min(k3, 0xFFFFFFFBu) : 0xFFFFFFFFu;
                        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)refs.x), r1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)refs.y),
                                       r2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)refs.z), r3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)refs.w);
                        // 5-comparator sorting network on (key, ref) pairs, all scalar
                        uint32_t ka = k0, kb = k1, kc = k2, kd = k3, ra = r0, rb = r1, rc = r2, rd = r3;
#define KZ_CSWAP(x, y, rx_, ry_) do { const bool s_ = y < x; const uint32_t tx_ = s_ ? y : x, ty_ = s_ ? x : y, tr_ = s_ ? ry_ : rx_, ts_ = s_ ? rx_ : ry_; x = tx_; y = ty_; rx_ = tr_; ry_ = ts_; } while (0)
                        KZ_CSWAP(ka, kb, ra, rb); KZ_CSWAP(kc, kd, rc, rd); KZ_CSWAP(ka, kc, ra, rc); KZ_CSWAP(kb, kd, rb, rd); KZ_CSWAP(kb, kc, rb, rc);
#undef KZ_CSWAP
                        // push far to near, descend into the nearest; a lane's own key of a pushed child is key[low two bits of the sorted key]
                        auto pushEntry = [&](uint32_t ks, uint32_t ref) {
                            if (KEYS && sp < KZ_PACKET_KEYS) { const uint32_t i = ks & 3u; kst[sp * KZ_BLOCK] = i == 0 ? key[0] : (i == 1 ? key[1] : (i == 2 ? key[2] : key[3])); }
                            if (sp < 64) kzWriteLane(stk0, ref, sp); else kzWriteLane(stk1, ref, sp - 64);
                            ++sp;
                        };
                        if (kd != 0xFFFFFFFFu) pushEntry(kd, rd);
                        if (kc != 0xFFFFFFFFu) pushEntry(kc, rc);
                        if (kb != 0xFFFFFFFFu) pushEntry(kb, rb);
                        cur = ra; descended = true;
                    }
                    if (descended) continue;
                } else {
                    const uint32_t start = (cur & 0x7fffffffu) >> 3, cnt = (cur & 7u) + 1;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        float t, u, v; uint32_t g;
                        if (STATS && live) cn.tris++;
                        if (!triTest(T.tris + start + i, o, d, tmin, tmax, t, u, v, g)) continue;
                        if (!found || t < bt || (t == bt && g < bgid)) { found = true; bt = t; bu = u; bv = v; bgid = g; tmax = t; }
                    }
                }
                bool more = false;
                while (sp > 0) {
                    --sp;
                    if (KEYS && sp < KZ_PACKET_KEYS) {                    // does any lane still need this entry?
                        const uint32_t k = kst[sp * KZ_BLOCK];
                        if (__ballot(k != 0xFFFFFFFFu && (k & ~3u) <= __float_as_uint(tmax)) == 0ull) continue;
                    }
                    cur = (uint32_t)(sp < 64 ? __builtin_amdgcn_readlane((int)stk0, sp) : __builtin_amdgcn_readlane((int)stk1, sp - 64));
                    more = true;
                    break;
                }
                if (!more) break;
            }
        }
        if (have) kzStoreStream(&W.hit[slot], make_float4(bt, bu, bv, __uint_as_float(bgid)))        /* no hit: still +inf, 0, 0, 0 */;
        if (FIX && found && bgid - P.ilGidLo <= P.ilGidSpan) {              // (rare) may be a triangle of an invisible light: look
            const int li = lightOfGid(T, bgid);
            if (li >= 0 && !T.lights[li].primaryVisibility) {
                RawHit rh; rh.t = bt; rh.u = bu; rh.v = bv; rh.tri = 0; rh.gid = bgid;
                Its its; postIntersect<false>(T, rh, its);
                const V3 no = its.p + P.traceBias * d;
                kzStoreStream(&W.shA[slot], make_float4(no.x, no.y, no.z, KZ_INF));            // Ray3f(o, d): mint = Epsilon, maxt = inf
                kzStoreStream(&W.shB[slot], make_float4(d.x, d.y, d.z, KZ_EPSILON));
                fixQueue[atomicAdd(fixCount, 1u)] = slot;
            }
        }
    }
    if (STATS) wfStatsFlush(W.stats, cn, 0);
}

// ---- pixel beams for the camera rays: one traversal per PIXEL, triangle tests per SAMPLE ----------------------------------------
// The S camera rays of a pixel in a pass differ by sub-pixel jitter only; kz_wf_trace_packet lets every one of them repeat the same
// ~35 node steps (61 VALU wave-instructions per ray on C4). Here the node work is done ONCE per pixel, by one lane:
//   kz_wf_beam        one lane = one pixel of the pass. Every sample position of the pixel lies in [px, px+1] x [py, py+1], and for a pinhole
//                     camera with an affine sample -> near-plane map (KzParams.beamOk) the rays of the pixel are the rays of the pyramid
//                     from the pinhole through the pixel's corners: the BEAM. The lane walks the BVH4 with the beam's CENTRAL ray against
//                     boxes padded by rho * (distance of the box's farthest point from the pinhole), rho = the largest chord between the
//                     central and a corner unit direction: a ray of the beam at distance s from the pinhole is within rho * s of the central
//                     ray's point at the same distance, so the central ray touches the padded box of every box a ray of the beam touches,
//                     and enters it no later. It writes the LEAVES it reaches, nearest first, with that entry distance as a lower bound
//                     of the ray parameter, until the list (KZ_BEAM_CAP) or its stack (KZ_BEAM_STACK) is full, and t_valid = the smallest
//                     bound of anything it left unexplored (infinity when it saw everything): every leaf that a ray of the beam can enter
//                     before t_valid is on the list.
//   kz_wf_trace_list  one lane = one camera ray: runs Mesh::rayIntersect (triTest, unchanged) on the triangles of the pixel's leaves in
//                     list order, skipping a leaf whose bound is behind the lane's closest hit so far. A hit in front of t_valid IS the
//                     closest hit (a nearer one would lie in a leaf that begins before it, hence on the list), and so is "no hit" when
//                     nothing was left unexplored; (t, u, v, triangle) come from the same function with the same tie rule: bit-identical
//                     to kz_wf_trace / kz_wf_trace_packet. The other rays go to fbQueue for the packet kernel with what is known about them
//                     (nothing in front of t_valid: tmin; the hit found so far: tmax).
// The lists depend on the pixels only: a pass context keeps them for the pixel chunk it last built them for.
#ifndef KZ_BEAM_STACK
#define KZ_BEAM_STACK 16              // open entries (ref + key) per beam in LDS. 32 until round 4: 64 KB per workgroup = 2 waves per SIMD; 16 = 5 waves per SIMD and half the
#endif                                // scan: kz_wf_beam 10.6 -> 6.4 ms on the C4 frame with the same lists downstream (list 2.8 ms, packet 0.5 ms per pass); 12: the packet kernel doubles, 8: x 16 (profiles/r04h_beam)
#define KZ_BEAM_UNBUILT 0xFFFFFFFFu   // head count of a pixel whose list has not been built
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_beam(KzParams P, KzDevTables T, const uint32_t *__restrict__ pixList, uint32_t nPix, int LS,
                                                       uint2 *__restrict__ entries, uint2 *__restrict__ heads) {
    // BEST-FIRST order: the open entries (child ref + entry distance of its padded box) of a lane form an unsorted set in LDS
    // ([LS][KZ_BLOCK] refs, then keys); the lane always takes the NEAREST one next, so the leaves reach the list in the order of their
    // bound and "everything unexplored begins behind the last key taken" holds at any time. A set that is full drops its farthest entry
    // (t_valid then ends there at the latest).
    extern __shared__ uint32_t s_stack[];
    uint32_t *stk = s_stack + threadIdx.x, *kst = s_stack + LS * KZ_BLOCK + threadIdx.x;
    const uint32_t pl = blockIdx.x * KZ_BLOCK + threadIdx.x;
    const uint32_t root = P.rootRef4;
    const uint32_t pxy = pl < nPix ? pixList[pl] : 0u;
    // The lists live per pixel of the FRAME (index y * width + x), whatever tile set or pixel chunk the pixel is rendered in: a pixel's list is built
    // once per replica (the camera belongs to the scene) - a head count of KZ_BEAM_UNBUILT marks a pixel nobody has built yet.
    const uint32_t fpix = (pxy >> 16) * (uint32_t)P.width + (pxy & 0xffffu);
    const bool todo = pl < nPix && heads[fpix].x == KZ_BEAM_UNBUILT;
    bool active = todo && root != 0xFFFFFFFFu;
    const float fx = (float)(pxy & 0xffffu), fy = (float)(pxy >> 16);
    const V3 O = mk(P.beamO[0], P.beamO[1], P.beamO[2]), U = mk(P.beamU[0], P.beamU[1], P.beamU[2]), V = mk(P.beamV[0], P.beamV[1], P.beamV[2]);
    // unit directions through the pixel's centre and corners (world axes; nearP = A + sx U + sy V)
    const V3 c00 = mk(P.beamA[0], P.beamA[1], P.beamA[2]) + fx * U + fy * V;
    const V3 uc = normalized(c00 + 0.5f * U + 0.5f * V);
    float rho = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const V3 e = normalized(c00 + (float)(k & 1) * U + (float)(k >> 1) * V) - uc; rho = fmaxf(rho, sqrtf(dot(e, e))); }
    rho = rho * 1.01f + 2e-6f;                            // + the rounding of a camera ray's direction (a few 1e-7) and of the corners
    // slab constants of the central ray (a zero component gets a tiny signed stand-in, as in kz_wf_trace)
    const float rx = 1.0f / (fabsf(uc.x) < 1e-20f ? copysignf(1e-20f, uc.x) : uc.x), ry = 1.0f / (fabsf(uc.y) < 1e-20f ? copysignf(1e-20f, uc.y) : uc.y),
                rz = 1.0f / (fabsf(uc.z) < 1e-20f ? copysignf(1e-20f, uc.z) : uc.z);
    float tvalid = KZ_INF;                                // distance from the pinhole before which nothing is left unexplored
    uint32_t cur = root, count = 0; float curKey = 0.f; int n = 0;
    uint2 *myList = entries + (size_t)fpix * KZ_BEAM_CAP;
    auto push = [&](uint32_t ref, uint32_t keyBits) {
        if (n < LS) { stk[n * KZ_BLOCK] = ref; kst[n * KZ_BLOCK] = keyBits; ++n; return; }
        int im = 0; uint32_t km = kst[0];                 // full: the farthest of the set and the newcomer stays out
        for (int i = 1; i < LS; ++i) { const uint32_t k = kst[i * KZ_BLOCK]; if (k > km) { km = k; im = i; } }
        if (keyBits < km) { stk[im * KZ_BLOCK] = ref; kst[im * KZ_BLOCK] = keyBits; tvalid = fminf(tvalid, __uint_as_float(km)); }
        else tvalid = fminf(tvalid, __uint_as_float(keyBits));
    };
    auto popMin = [&]() -> bool {
        if (n == 0) return false;
        int im = 0; uint32_t km = kst[0];
        for (int i = 1; i < n; ++i) { const uint32_t k = kst[i * KZ_BLOCK]; if (k < km) { km = k; im = i; } }
        if (!(__uint_as_float(km) < tvalid)) return false;                     // what begins behind t_valid cannot be made use of
        cur = stk[im * KZ_BLOCK]; curKey = __uint_as_float(km);
        --n;
        if (im != n) { stk[im * KZ_BLOCK] = stk[n * KZ_BLOCK]; kst[im * KZ_BLOCK] = kst[n * KZ_BLOCK]; }
        return true;
    };
    // One loop, one step per iteration for every lane that still works on its pixel: a lane that holds an inner node tests its four children, a lane that
    // holds a leaf writes it to the list, and both take the nearest open entry next. (Until round 4 this was a while-while loop - all lanes descend, then all
    // lanes at a leaf append - but a best-first walk reaches a leaf every few steps, each lane at another time, and the lanes spent their time waiting for
    // one another: 19 of 64 busy.)
    while (__any(active)) {
        bool needPop = false;
        if (active && !(cur & 0x80000000u)) {
            const uint4 *np = kzNode4Ptr(T, cur);
            const uint4 q0 = np[0], q1 = np[1], q2 = np[2], refs = np[3];
            const float sX = __uint_as_float(q0.w), sY = __uint_as_float(q2.z), sZ = __uint_as_float(q2.w);
            const float pX = __uint_as_float(q0.x) - O.x, pY = __uint_as_float(q0.y) - O.y, pZ = __uint_as_float(q0.z) - O.z;     // box coordinates relative to the pinhole
            const uint32_t r[4] = {refs.x, refs.y, refs.z, refs.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float qlx = (float)((q1.x >> (8 * i)) & 0xffu), qly = (float)((q1.y >> (8 * i)) & 0xffu), qlz = (float)((q1.z >> (8 * i)) & 0xffu);
                const float qhx = (float)((q1.w >> (8 * i)) & 0xffu), qhy = (float)((q2.x >> (8 * i)) & 0xffu), qhz = (float)((q2.y >> (8 * i)) & 0xffu);
                const float lx = fmaf(qlx, sX, pX), ly = fmaf(qly, sY, pY), lz = fmaf(qlz, sZ, pZ), hx = fmaf(qhx, sX, pX), hy = fmaf(qhy, sY, pY), hz = fmaf(qhz, sZ, pZ);
                // pad: rho x an upper bound of the distance of the box's farthest point, + 2^-19 of the coordinates (the quantised planes
                // are conservative for the kernels' own slab expression; this form rounds differently)
                const float reach = fmaxf(fabsf(lx), fabsf(hx)) + fmaxf(fabsf(ly), fabsf(hy)) + fmaxf(fabsf(lz), fabsf(hz));
                const float pad = rho * reach + 1.9e-6f * (reach + fabsf(O.x) + fabsf(O.y) + fabsf(O.z));
                const float t0x = (lx - pad) * rx, t1x = (hx + pad) * rx, t0y = (ly - pad) * ry, t1y = (hy + pad) * ry, t0z = (lz - pad) * rz, t1z = (hz + pad) * rz;
                const float nn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.f));
                const float ff = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z)) * 1.0000004f;
                // a child's bound is not below its parent's (its padded box lies inside the parent's; the max guards the rounding)
                const float key = fmaxf(nn * 0.999999f, curKey);
                if (qhx >= qlx && nn <= ff && key < tvalid) push(r[i], __float_as_uint(key));      // (an empty slot has qlo = 255 > qhi = 0)
            }
            needPop = true;
        } else if (active) {
            if (count < KZ_BEAM_CAP) { myList[count] = make_uint2(cur, __float_as_uint(curKey)); ++count; needPop = true; }
            else { tvalid = fminf(tvalid, curKey); active = false; }      // the list is full: this leaf and all that is open begin no nearer
        }
        if (needPop && !popMin()) active = false;
    }
    if (todo) heads[fpix] = make_uint2(count, __float_as_uint(tvalid));      // (distances from the pinhole: kz_wf_trace_list scales its ray parameters by |d|)
}

// stats only: pixels with a list, list entries, pixels whose list is complete (t_valid = infinity)
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_beam_count(const uint2 *__restrict__ heads, const uint32_t *__restrict__ pixList, int width, uint32_t nPix, unsigned long long *__restrict__ out) {
    const uint32_t pl = blockIdx.x * KZ_BLOCK + threadIdx.x;
    uint2 h = make_uint2(0u, 0u);
    if (pl < nPix) { const uint32_t pxy = pixList[pl]; h = heads[(pxy >> 16) * (uint32_t)width + (pxy & 0xffffu)]; }
    unsigned long long v[3] = {pl < nPix ? 1ull : 0ull, (unsigned long long)h.x, (pl < nPix && !(__uint_as_float(h.y) < KZ_INF)) ? 1ull : 0ull};
    for (int k = 0; k < 3; ++k) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(&out[k], x);
    }
}

// closest hit of the camera rays from the pixel lists; rays the lists cannot decide go to fbQueue (kz_wf_trace_packet takes them)
template <bool STATS, bool FIX>
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_trace_list(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ pixList, uint32_t nItems, uint32_t S, const uint2 *__restrict__ entries,
                                                             const uint2 *__restrict__ heads, uint32_t *__restrict__ fbQueue, uint32_t *__restrict__ fbCount,
                                                             uint32_t *__restrict__ fixQueue, uint32_t *__restrict__ fixCount) {
    const uint32_t slot = blockIdx.x * KZ_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    Counters cn = {0, 0, 0, 0, 0, 0};
    const bool have = slot < nItems;
    bool undecided = false;
    if (have) {
        const uint32_t pl = (S & 63u) == 0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)slot) / S : slot / S;      // (a wave of 64 | S samples is one pixel: scalar division)
        const float4 a = kzLoadStream(&W.rayA[slot]), b = kzLoadStream(&W.rayB[slot]);
        const V3 o = mk(a.x, a.y, a.z), d = mk(b.x, b.y, b.z);
        const float tmin = a.w;
        float tmax = b.w;
        bool found = false; float bt = KZ_INF, bu = 0.f, bv = 0.f; uint32_t bgid = 0;
        const bool finite = rayIsFinite(o, d);
        const float len = sqrtf(dot(d, d)), lenUp = len * 1.000002f;
        float tvalidDist = KZ_INF;                                             // the list's bounds are DISTANCES from the pinhole: parameter x |d|
        auto consider = [&](float t, float u, float v, uint32_t g) { if (!found || t < bt || (t == bt && g < bgid)) { found = true; bt = t; bu = u; bv = v; bgid = g; tmax = t; } };
        // per-lane lists (a wave that spans pixels)
        auto walk = [&](const uint2 *__restrict__ lst, const uint2 head) {
            tvalidDist = __uint_as_float(head.y);
            if (!finite) return;
            for (uint32_t j = 0; j < head.x; ++j) {
                const uint2 e = lst[j];
                if (__uint_as_float(e.y) > tmax * lenUp) continue;         // the leaf begins behind the closest hit so far
                const uint32_t start = (e.x & 0x7fffffffu) >> 3, cnt = (e.x & 7u) + 1;
                for (uint32_t i = 0; i < cnt; ++i) {
                    float t, u, v; uint32_t g;
                    if (STATS) cn.tris++;
                    if (triTest(T.tris + start + i, o, d, tmin, tmax, t, u, v, g)) consider(t, u, v, g);
                }
            }
        };
        // ONE list for the wave: entries and triangles arrive through the scalar cache, and every one of those loads is a dependent round trip the
        // whole wave waits for. So the next entry is fetched while this one is tested, and the triangles two at a time. The list is in the order of the
        // beam's best-first walk - its bounds never decrease: when a leaf begins behind the closest hit of EVERY ray, so do all the leaves after it.
        auto walkShared = [&](const uint2 *__restrict__ lst, const uint2 head) {
            tvalidDist = __uint_as_float(head.y);
            if (!finite || head.x == 0) return;
            uint2 e = lst[0];
            for (uint32_t j = 0; j < head.x; ++j) {
                const uint2 eNext = lst[min(j + 1u, head.x - 1u)];
                const bool inFront = __uint_as_float(e.y) <= tmax * lenUp;
                if (!__any(inFront)) break;
                if (inFront) {
                    const uint32_t start = (e.x & 0x7fffffffu) >> 3, cnt = (e.x & 7u) + 1;
                    for (uint32_t i = 0; i < cnt; i += 2) {
                        const float4 *tp = reinterpret_cast<const float4 *>(T.tris + start + i), *tq = tp + (i + 1 < cnt ? 3 : 0);
                        const float4 a0 = tp[0], b0 = tp[1], c0 = tp[2], a1 = tq[0], b1 = tq[1], c1 = tq[2];
                        float t, u, v; uint32_t g;
                        if (STATS) cn.tris++;
                        if (triTestV(a0, b0, c0, o, d, tmin, tmax, t, u, v, g)) consider(t, u, v, g);
                        if (i + 1 < cnt) {
                            if (STATS) cn.tris++;
                            if (triTestV(a1, b1, c1, o, d, tmin, tmax, t, u, v, g)) consider(t, u, v, g);
                        }
                    }
                }
                e = eNext;
            }
        };
        // When the 64 rays of the wave belong to ONE pixel (S a multiple of 64: every default), the list is wave-uniform: its address comes from the
        // first lane's pixel (an SGPR), so the head and the entries arrive through the scalar cache and an entry's load no longer queues behind the
        // lanes' vector loads of the triangle before it. Waves that span pixels walk per-lane lists.
        const uint32_t plU = (uint32_t)__builtin_amdgcn_readfirstlane((int)pl);
        if (__all(pl == plU)) { const uint32_t pxy = pixList[plU], fp = (pxy >> 16) * (uint32_t)P.width + (pxy & 0xffffu); walkShared(entries + (size_t)fp * KZ_BEAM_CAP, heads[fp]); }
        else { const uint32_t pxy = pixList[pl], fp = (pxy >> 16) * (uint32_t)P.width + (pxy & 0xffffu); walk(entries + (size_t)fp * KZ_BEAM_CAP, heads[fp]); }
        // decided: a hit in front of everything unexplored, or nothing unexplored at all (a non-finite ray hits nothing)
        undecided = finite && !(found ? bt * lenUp < tvalidDist : !(tvalidDist < KZ_INF));
        if (STATS && !undecided) cn.rays++;                                    // (an undecided ray is counted by the kernel that decides it)
        if (undecided) {
            // what the list has established travels with the ray: nothing is hit in front of t_valid, and nothing behind the hit found matters
            kzStoreStream(&W.rayA[slot], make_float4(a.x, a.y, a.z, fmaxf(tmin, tvalidDist / lenUp * 0.999998f)));
            if (found) kzStoreStream(&W.rayB[slot], make_float4(b.x, b.y, b.z, bt));
        } else {
            kzStoreStream(&W.hit[slot], make_float4(bt, bu, bv, __uint_as_float(bgid)))        /* no hit: still +inf, 0, 0, 0 */;
            if (FIX && found && bgid - P.ilGidLo <= P.ilGidSpan) {              // (rare) may be a triangle of an invisible light: see kz_wf_trace_packet
                const int li = lightOfGid(T, bgid);
                if (li >= 0 && !T.lights[li].primaryVisibility) {
                    RawHit rh; rh.t = bt; rh.u = bu; rh.v = bv; rh.tri = 0; rh.gid = bgid;
                    Its its; postIntersect<false>(T, rh, its);
                    const V3 no = its.p + P.traceBias * d;
                    kzStoreStream(&W.shA[slot], make_float4(no.x, no.y, no.z, KZ_INF));
                    kzStoreStream(&W.shB[slot], make_float4(d.x, d.y, d.z, KZ_EPSILON));
                    fixQueue[atomicAdd(fixCount, 1u)] = slot;
                }
            }
        }
    }
    {   // undecided rays: appended in slot order within the wave (neighbours in the queue stay neighbours in the image)
        const unsigned long long m = __ballot(undecided);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(fbCount, (uint32_t)__popcll(m));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (undecided) fbQueue[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = slot;
        }
    }
    if (STATS) wfStatsFlush(W.stats, cn, 0);
}

// ---- final: the ray after the last bounce contributes only the background on a miss (integrator.cpp:315-318) ------------
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_final(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ queue, const uint32_t *__restrict__ countPtr) {
    const uint32_t count = *countPtr;
    for (uint32_t qi = blockIdx.x * KZ_BLOCK + threadIdx.x; qi < count; qi += gridDim.x * KZ_BLOCK) {
        const uint32_t slot = queue[qi];
        const float4 h = kzLoadStream(&W.hit[slot]);
        if (h.x < KZ_INF) continue;
        const float4 rb = kzLoadStream(&W.rayB[slot]), th = kzLoadStream(&W.thr[slot]);
        const V3 bg = backgroundRadiance(P, T, mk(rb.x, rb.y, rb.z));
        unsafeAtomicAdd(W.outR + slot, th.x * bg.x); unsafeAtomicAdd(W.outG + slot, th.y * bg.y); unsafeAtomicAdd(W.outB + slot, th.z * bg.z);
    }
}

// ---- stats only: samples rendered and invalid radiance values (Color3f::isValid, common.cpp:384-391) --------------------
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_count(KzWf W, uint32_t nItems) {
    Counters cn = {0, 0, 0, 0, 0, 0};
    const uint32_t i = blockIdx.x * KZ_BLOCK + threadIdx.x;
    if (i < nItems) {
        const float r = W.outR[i], g = W.outG[i], b = W.outB[i];
        if (!(r >= 0 && g >= 0 && b >= 0 && isfinite(r) && isfinite(g) && isfinite(b))) cn.dropped++;
    }
    wfStatsFlush(W.stats, cn, i < nItems ? 1u : 0u);
}
