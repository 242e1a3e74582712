// kz_experiments.h — kernels of experiments that were built, measured on MI355X and NOT adopted (profiles/r01*, r02*).
// Compiled only with -DKZ_EXPERIMENTS (scripts/build_variant.sh experiments -DKZ_EXPERIMENTS); the default library does not contain them.
// They are kept because each one is held bit-identical to the product kernels by tests/test_gpu_configs.py (which runs those cases
// only against a library that reports KZ_BUILD_EXPERIMENTS in kz_build_flags()):
//   kz_wf_extend / kz_wf_shadow   the non-persistent round-1 traversal launches (KzTuning.legacyTrace)
//   kz_wf_trace_x                 the round-2 per-lane traversal kernel with every option it carried: BVH2 (WIDE = false), per-lane key
//                                 stack (KEYS), LDS top-of-tree (TOP), mixed closest-hit + shadow launches (MODE 3), full sibling sort
//   kz_wf_trace_dq                decoupled leaf phase (per-wave LDS job queue)
#pragma once
#include "../../kz_wavefront.h"

// ---- extend: closest hit for the rays of a queue (queue == nullptr: identity over [0, count)) --------------------------
// KEEP: leave the previous hit record in place on a miss and read the ray from the shA/shB pair (walk-through, H6).
template <bool STATS, bool KEEP>
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_extend(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ queue,
                                                         const uint32_t *__restrict__ countPtr, uint32_t countImm) {
    extern __shared__ uint32_t s_stack[];
    const uint32_t count = countPtr ? *countPtr : countImm;
    Counters cn = {0, 0, 0, 0, 0, 0};
    for (uint32_t base = blockIdx.x * KZ_BLOCK; base < count; base += gridDim.x * KZ_BLOCK) {
        const uint32_t qi = base + threadIdx.x;
        if (qi >= count) continue;
        const uint32_t slot = queue ? queue[qi] : qi;
        float4 a, b;
        if (KEEP) { const float4 sa = W.shA[slot], sb = W.shB[slot]; a = make_float4(sa.x, sa.y, sa.z, sb.w); b = make_float4(sb.x, sb.y, sb.z, sa.w); }
        else { a = W.rayA[slot]; b = W.rayB[slot]; }
        RawHit rh;
        const bool found = closestHit<STATS>(T, P.rootRef, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), a.w, b.w, rh, s_stack + threadIdx.x, cn);
        if (found) W.hit[slot] = make_float4(rh.t, rh.u, rh.v, __uint_as_float(rh.gid));
        else if (!KEEP) W.hit[slot] = make_float4(KZ_INF, 0.f, 0.f, 0.f);
    }
    if (STATS) wfStatsFlush(W.stats, cn, 0);
}

// ---- shadow(iter): occlusion test, adds the pending radiance (integrator.cpp:257-295) -----------------------------------
template <bool STATS>
__global__ __launch_bounds__(KZ_BLOCK) void kz_wf_shadow(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ queue,
                                                         const uint32_t *__restrict__ countPtr) {
    extern __shared__ uint32_t s_stack[];
    const uint32_t count = *countPtr;
    Counters cn = {0, 0, 0, 0, 0, 0};
    for (uint32_t base = blockIdx.x * KZ_BLOCK; base < count; base += gridDim.x * KZ_BLOCK) {
        const uint32_t qi = base + threadIdx.x;
        if (qi >= count) continue;
        const uint32_t slot = queue[qi];
        const float4 a = W.shA[slot], b = W.shB[slot];
        const bool occluded = shadowOccluded<STATS>(P, T, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), b.w, a.w, s_stack + threadIdx.x, cn);
        if (!occluded) {
            const float4 l = W.shL[slot];
            W.outR[slot] += l.x; W.outG[slot] += l.y; W.outB[slot] += l.z;
        }
    }
    if (STATS) wfStatsFlush(W.stats, cn, 0);
}

template <int MODE, bool STATS, bool WIDE, bool KEYS = false, bool TOP = false>
__global__ __launch_bounds__(KZ_BLOCK) __attribute__((amdgpu_waves_per_eu(KZ_TRACE_WAVES, KZ_TRACE_WAVES))) void kz_wf_trace_x(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ queue,
                                                        const uint32_t *__restrict__ countPtr, uint32_t countImm, uint32_t *__restrict__ head, KzTune tune,
                                                        const uint32_t *__restrict__ queueB, const uint32_t *__restrict__ countPtrB) {
    extern __shared__ uint32_t s_stack[];
    uint32_t *stk = s_stack + threadIdx.x;
    const uint32_t countA = countPtr ? *countPtr : countImm;
    const uint32_t count = countA + ((MODE == 3) ? *countPtrB : 0u);
    // MODE 4 = MODE 2 without the walk-through machinery: a shadow ray whose segment crosses a triangle of an invisible light (rare) is
    // not traced here but appended to the queue passed in queueB / countPtrB, which a MODE 2 launch takes afterwards. With `literal`
    // constant false the closest-hit bookkeeping of the shadow lanes (hit distance, barycentrics, triangle ids) disappears from this
    // instantiation: 56 VGPRs and no spills instead of 64 with 9 spilled (shadow stage 23.9 -> 22.6 ms on C4 with the test switched off).
    constexpr bool SHADOW = MODE == 2 || MODE == 4;
    int kind = (MODE == 3) ? 0 : (MODE == 4 ? 2 : MODE);              // per-lane ray kind; a compile-time constant unless the launch is mixed
    const int lane = threadIdx.x & 63;
    const int LS = tune.ldsStack;
    // overflow rows live at tune.ovf[row * ovfStride + thread]: the (rare) deep case forms its address from the scalar base and a 32-bit
    // thread index instead of keeping a 64-bit per-lane pointer alive through the loop
    uint32_t *const ovfBase = tune.ovf;
    const uint32_t ovfLane = blockIdx.x * KZ_BLOCK + threadIdx.x;
    const size_t ovfStride = tune.ovfStride;
#define ovf(row) ovfBase[(size_t)(row) * ovfStride + ovfLane]
    // KEYS (closest-hit rays on the BVH4): every stack entry carries the entry distance of its box (the sortable key of node4Keys), in a
    // second LDS column block / in the odd rows of the overflow area. An entry whose box starts behind the closest hit found so far
    // is dropped at pop time for the price of two LDS reads and a compare instead of a full node step on four boxes that all miss.
    const int kOff = (LS + 1) * KZ_BLOCK;              // key column block behind the ref column block (KEYS only)
    // TOP (tune.ldsTop > 0; north_star: "LDS-staged BVH node packets"): the first ldsTop packets of the breadth-first BVH4 array = the top
    // levels of the tree are copied into LDS behind the stacks by the whole workgroup and node steps on them read LDS instead of L1.
    // Measured on C4 (profiles/r02c_lds_top): 13 % fewer L1 accesses, 1-3 % less time, and the test in the node step costs the
    // kernels that do not use it 5 % -> a separate instantiation, off by default.
    const uint32_t nTop = (WIDE && TOP) ? (uint32_t)tune.ldsTop : 0u;
    const uint4 *s_top = reinterpret_cast<const uint4 *>(s_stack + (LS + 1) * KZ_BLOCK * (KEYS ? 2 : 1));
    if (nTop) {
        uint4 *w = reinterpret_cast<uint4 *>(s_stack + (LS + 1) * KZ_BLOCK * (KEYS ? 2 : 1));
        const uint4 *src = reinterpret_cast<const uint4 *>(T.nodes4);
        for (uint32_t i = threadIdx.x; i < nTop * 4u; i += KZ_BLOCK) w[i] = src[i];
        __syncthreads();
    }
    const size_t ovfW = KEYS ? 2 : 1;
    auto push = [&](int &sp_, uint32_t v, uint32_t k) {
        if (sp_ < LS) { stk[sp_ * KZ_BLOCK] = v; if (KEYS) stk[kOff + sp_ * KZ_BLOCK] = k; }
        else { ovf((size_t)(sp_ - LS) * ovfW) = v; if (KEYS) ovf((size_t)(sp_ - LS) * 2 + 1) = k; }
        ++sp_;
    };
    const uint32_t root = WIDE ? P.rootRef4 : P.rootRef;
    const float eps = P.traceBias;
    Counters cn = {0, 0, 0, 0, 0, 0};
    // Queue entries are claimed in batches. The FIRST batch of a wave is static (wave w owns entries [w*batch, (w+1)*batch)): a
    // launch on a short queue then costs no atomics at all, where 8192 waves hitting one counter took ~95 us (one word serves ~88
    // dequeues/us) - the whole duration of the late, nearly empty bounces of a small frame. Further batches come from the shared counter.
    const uint32_t nWaves = gridDim.x * (KZ_BLOCK / 64), waveId = blockIdx.x * (KZ_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t batch = max(64u, min((uint32_t)tune.batch, ((count + nWaves - 1) / nWaves + 63u) & ~63u));   // short queues: spread over all waves
    const bool staticOnly = (unsigned long long)nWaves * batch >= count;
    uint32_t poolNext = min(waveId * batch, count), poolEnd = min(poolNext + batch, count);
    bool exhausted = false;
    bool active = false, literal = false;
    V3 o = mk(0.f), d = mk(0.f);
    float rx = 0.f, ry = 0.f, rz = 0.f, tmin = 0.f, tmax = 0.f, segMax = 0.f;
    uint32_t cur = 0, slot = 0; int sp = 0;
    bool found = false; float bt = 0.f, bu = 0.f, bv = 0.f; uint32_t btri = 0, bgid = 0;

    // next stack entry that can still matter -> cur; false when the stack is empty
    auto popNext = [&]() -> bool {
        while (sp > 0) {
            --sp;
            // (LDS reads of the entry or of the scratch row, replaced by the global entry in the rare deep case: a select between an LDS and a
            // global address would become a generic-pointer load)
            const int row = min(sp, LS) * KZ_BLOCK;
            uint32_t v = stk[row], k = KEYS ? stk[kOff + row] : 0u;
            if (sp >= LS) { v = ovf((size_t)(sp - LS) * ovfW); if (KEYS) k = ovf((size_t)(sp - LS) * 2 + 1); }
            if (!KEYS || (k & ~3u) <= __float_as_uint(tmax)) { cur = v; return true; }      // key = bits of max(tnear, tmin), low two bits = slot
        }
        return false;
    };
#ifdef KZ_LANESTAT
    // development build only (-DKZ_LANESTAT): where the lanes of the while-while loop are, summed per wave (wave-uniform counts)
    unsigned long long lsNodeIters = 0, lsActiveAtNode = 0, lsInnerAtNode = 0, lsLeafPhases = 0, lsLeafLanes = 0, lsRefills = 0, lsRefillLanes = 0, lsTriIters = 0;
#endif
    auto addPending = [&]() { const float4 l = W.shL[slot]; W.outR[slot] += l.x; W.outG[slot] += l.y; W.outB[slot] += l.z; };
    // the lane's stack ran empty: publish the result (or, for a literal shadow lane, decide / walk through the light)
    auto finish = [&]() {
        active = false;
        if (kind == 0) W.hit[slot] = found ? make_float4(bt, bu, bv, __uint_as_float(bgid)) : make_float4(KZ_INF, 0.f, 0.f, 0.f);
        if (kind == 1) { if (found) W.hit[slot] = make_float4(bt, bu, bv, __uint_as_float(bgid)); }
        if (kind == 2) {
            if (MODE == 4 || !literal || !found) addPending();                       // nothing on the segment
            else {
                const uint32_t om = __float_as_uint(reinterpret_cast<const float4 *>(T.tris + btri)[2].y);      // (the leaf triangle is in cache; its shading record is not)
                const int ol = T.meshes[om].light;
                if (ol >= 0 && !T.lights[ol].primaryVisibility) {                     // walk through (integrator.cpp:273-274)
                    o = o + d * (bt + eps); tmin = eps; segMax = segMax - bt; tmax = segMax;
                    found = false; bt = KZ_INF; cur = root; sp = 0; active = true;
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
                    if (lane == 0) b = atomicAdd(head, batch);
                    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    b = (b < 0xFFFFFFFFu - nWaves * batch) ? b + nWaves * batch : count;            // dynamic batches start behind the static ones
                }
                if (b >= count) { exhausted = true; poolNext = poolEnd = 0; }
                else { poolNext = b; poolEnd = min(b + batch, count); }
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
                    if (MODE == 3) { kind = qi < countA ? 0 : 2; slot = qi < countA ? queue[qi] : queueB[qi - countA]; }
                    else slot = queue ? queue[qi] : qi;
                    float4 a, b;
                    if (kind == 0) { a = W.rayA[slot]; b = W.rayB[slot]; }
                    else { const float4 sa = W.shA[slot], sb = W.shB[slot]; a = make_float4(sa.x, sa.y, sa.z, sb.w); b = make_float4(sb.x, sb.y, sb.z, sa.w); }
                    o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z); tmin = a.w; tmax = b.w; segMax = b.w;
                    found = false; bt = KZ_INF; bu = bv = 0.f; btri = 0; bgid = 0; literal = false;
                    if (STATS && MODE != 4) cn.rays++;
                    if ((root != 0xFFFFFFFFu) && rayIsFinite(o, d)) {
                        if (WIDE) {
                            // The FMA slab form q*(s*rcp) + (p-o)*rcp turns into inf - inf = NaN for a zero direction component,
                            // which would switch that axis off (a huge slab of the tree gets walked). A tiny signed stand-in keeps
                            // every product finite; the sign of (box - origin) * 1e20 still decides the slab exactly as 1/0 would.
                            rx = rcpExact(fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x);
                            ry = rcpExact(fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y);
                            rz = rcpExact(fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z);
                        } else { rx = rcpExact(d.x); ry = rcpExact(d.y); rz = rcpExact(d.z); }
                        cur = root; sp = 0; active = true;
                        if (MODE == 4) {
                            if (invisibleLightOnSegment(P, T, o, d, rx, ry, rz, tmin, tmax)) {       // (launched only when P.shadowFast)
                                const_cast<uint32_t *>(queueB)[atomicAdd(const_cast<uint32_t *>(countPtrB), 1u)] = slot;
                                active = false;
                            } else if (STATS) cn.rays++;
                        } else if (kind == 2) literal = !P.shadowFast || invisibleLightOnSegment(P, T, o, d, rx, ry, rz, tmin, tmax);
                    } else {
                        // a ray that cannot hit anything (empty scene, non-finite origin/direction)
                        if (kind == 0) W.hit[slot] = make_float4(KZ_INF, 0.f, 0.f, 0.f);
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
                bool empty = false;
                if (STATS) cn.nodes++;
#ifndef KZ_TRAV_ORDER
#define KZ_TRAV_ORDER 1
#endif
#ifndef KZ_SHADOW_SLOT_ORDER
#define KZ_SHADOW_SLOT_ORDER 1
#endif
                if (WIDE && KZ_TRAV_ORDER) {
                    // Child order. Closest-hit rays descend into the NEAREST hit child and leave the other hit children on the
                    // stack in slot order (a full sort of the siblings buys 6 % fewer node visits and costs 12 % more
                    // instructions per visit); any-hit shadow rays take the hit children in slot order altogether. Pushes
                    // are branch-free: a hit child lands on the next free slot, a missed one on the lane's scratch slot.
                    uint32_t key[4]; uint4 refs;
                    bool p0, p1, p2, p3;                                          // child i goes on the stack
                    uint32_t nxt; bool any;
                    constexpr bool ORDERED = !(SHADOW && KZ_SHADOW_SLOT_ORDER);      // shadow launches take the hit children in slot order
                    if (TOP) {
                        uint4 q0, q1, q2;
                        if (cur < nTop) { const uint4 *lp = s_top + cur * 4u; q0 = lp[0]; q1 = lp[1]; q2 = lp[2]; refs = lp[3]; }
                        else { const uint4 *np = reinterpret_cast<const uint4 *>(T.nodes4 + cur); q0 = np[0]; q1 = np[1]; q2 = np[2]; refs = np[3]; }
                        node4KeysOf<ORDERED>(q0, q1, q2, o, rx, ry, rz, tmin, tmax, key);
                    } else node4Keys<ORDERED>(T, cur, o, rx, ry, rz, tmin, tmax, key, refs);
                    if (SHADOW && KZ_SHADOW_SLOT_ORDER) {
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
                    const int c1 = (int)p0, c2 = c1 + (int)p1, c3 = c2 + (int)p2, np = c3 + (int)p3;
                    if (sp + 3 <= LS) {                                              // common case: everything stays in LDS
                        const int o0 = (p0 ? sp : LS) * KZ_BLOCK, o1 = (p1 ? sp + c1 : LS) * KZ_BLOCK, o2 = (p2 ? sp + c2 : LS) * KZ_BLOCK, o3 = (p3 ? sp + c3 : LS) * KZ_BLOCK;
                        if (!(SHADOW && KZ_SHADOW_SLOT_ORDER)) { stk[o0] = refs.x; if (KEYS) stk[kOff + o0] = key[0]; }
                        stk[o1] = refs.y; stk[o2] = refs.z; stk[o3] = refs.w;
                        if (KEYS) { stk[kOff + o1] = key[1]; stk[kOff + o2] = key[2]; stk[kOff + o3] = key[3]; }
                        sp += np;
                    } else {
                        if (p0) push(sp, refs.x, key[0]);
                        if (p1) push(sp, refs.y, key[1]);
                        if (p2) push(sp, refs.z, key[2]);
                        if (p3) push(sp, refs.w, key[3]);
                    }
                    if (any) cur = nxt; else empty = true;
                } else if (WIDE) {
                    const Node4Test nt = node4Test(T, cur, o, rx, ry, rz, tmin, tmax);
                    // sorted keys: misses (0xFFFFFFFF) come last, so the hit children are k0 .. k(h-1). The nearest becomes
                    // `cur`; the others go on the stack far-to-near WITHOUT branches: child j (1..3) lands on slot
                    // sp + np - j when it was hit and on the lane's scratch slot otherwise (np = number of pushes).
                    const uint32_t c0 = pick4b(nt.refs, nt.k0), c1 = pick4b(nt.refs, nt.k1), c2 = pick4b(nt.refs, nt.k2), c3 = pick4b(nt.refs, nt.k3);
                    const int h1 = nt.k1 != 0xFFFFFFFFu, h2 = nt.k2 != 0xFFFFFFFFu, h3 = nt.k3 != 0xFFFFFFFFu;
                    const int np = h1 + h2 + h3;
                    if (sp + 3 <= LS) {                                              // common case: everything stays in LDS
                        const int o1 = (h1 ? sp + np - 1 : LS) * KZ_BLOCK, o2 = (h2 ? sp + np - 2 : LS) * KZ_BLOCK, o3 = (h3 ? sp : LS) * KZ_BLOCK;
                        stk[o1] = c1; stk[o2] = c2; stk[o3] = c3;
                        if (KEYS) { stk[kOff + o1] = nt.k1; stk[kOff + o2] = nt.k2; stk[kOff + o3] = nt.k3; }
                        sp += np;
                    } else {
                        if (h3) push(sp, c3, nt.k3);
                        if (h2) push(sp, c2, nt.k2);
                        if (h1) push(sp, c1, nt.k1);
                    }
                    if (nt.k0 != 0xFFFFFFFFu) cur = c0; else empty = true;
                } else {
                    const NodeTest nt = nodeTest(T, cur, o, rx, ry, rz, tmin, tmax);
                    if (nt.h0 && nt.h1) {
                        const bool swap = nt.n1 < nt.n0;
                        push(sp, swap ? nt.c0 : nt.c1, 0u);
                        cur = swap ? nt.c1 : nt.c0;
                    } else if (nt.h0) cur = nt.c0;
                    else if (nt.h1) cur = nt.c1;
                    else empty = true;
                }
                if (empty) { if (!popNext()) finish(); }
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

// ---- traversal with a decoupled leaf phase (tune.leafQueue) --------------------------------------------------------------
// kz_wf_trace alternates a node phase and a leaf phase per wave; the lane statistics of the bounce rays (-DKZ_LANESTAT) show what that
// costs: of 51 lanes that hold a ray, 37 step through a node while 14 wait with a leaf, and every 2.8 node steps a leaf phase runs
// with 22 lanes. Here a lane that reaches a leaf does not wait: it appends a (lane, leaf) JOB to a per-wave queue in LDS and goes on
// with its next stack entry; whenever 64 jobs are waiting (or no lane can descend) the whole wave runs them, one job per lane:
// the job lane fetches the OWNER's ray with ds_bpermute, runs Mesh::rayIntersect on the leaf's triangles and publishes a hit with a
// 64-bit LDS atomicMin on (bits of t, triangle id) - exactly the tie rule of the closest hit (lower id on equal t) - or, for an any-hit
// shadow ray, by raising the owner's flag. An owner picks up its current closest distance from LDS before every node step (a stale,
// larger tmax only costs visits, never a hit), and publishes its result once its stack is empty AND its last job has run.
// MODE 0: closest hit -> W.hit; MODE 2: any-hit shadow test -> adds W.shL (rays that would need the literal walk-through of an
// invisible light are not handled here: the host uses this kernel only when shadowFast holds and routes those rays to kz_wf_trace).
#define KZ_DQ_JOBS 128                       // ring capacity (a batch of 64 is run before a step could overflow it)
template <int MODE, bool STATS>
__global__ __launch_bounds__(KZ_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void kz_wf_trace_dq(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ queue,
                                                        const uint32_t *__restrict__ countPtr, uint32_t countImm, uint32_t *__restrict__ head, KzTune tune,
                                                        uint32_t *__restrict__ literalQueue, uint32_t *__restrict__ literalCount) {
    extern __shared__ uint32_t s_dq[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int LS = tune.ldsStack;
    // per-wave LDS: [LS + 1][64] stack | best[64] u64 | res[3][64] | jobLeaf[128] | jobOwner[128]
    const int waveWords = (LS + 1) * 64 + 128 + 192 + 2 * KZ_DQ_JOBS;
    uint32_t *const wbase = s_dq + wave * waveWords;
    uint32_t *const stk = wbase + lane;                                          // row stride 64
    unsigned long long *const best = reinterpret_cast<unsigned long long *>(wbase + (LS + 1) * 64);
    uint32_t *const res = wbase + (LS + 1) * 64 + 128;                           // [3][64]: triangle, u, v
    uint32_t *const jobLeaf = res + 192, *const jobOwner = jobLeaf + KZ_DQ_JOBS;
    uint32_t *const ovfBase = tune.ovf;
    const uint32_t ovfLane = blockIdx.x * KZ_BLOCK + threadIdx.x;
    const size_t ovfStride = tune.ovfStride;
    const uint32_t count = countPtr ? *countPtr : countImm;
    const uint32_t root = P.rootRef4;
    Counters cn = {0, 0, 0, 0, 0, 0};
    const uint32_t nWaves = gridDim.x * (KZ_BLOCK / 64), waveId = blockIdx.x * (KZ_BLOCK / 64) + (threadIdx.x >> 6);
    const uint32_t batch = max(64u, min((uint32_t)tune.batch, ((count + nWaves - 1) / nWaves + 63u) & ~63u));
    const bool staticOnly = (unsigned long long)nWaves * batch >= count;
    uint32_t poolNext = min(waveId * batch, count), poolEnd = min(poolNext + batch, count);
    bool exhausted = false;
    // lane state: 0 idle, 1 traversing (cur valid), 2 draining (stack empty, waiting for its last job)
    int state = 0;
    V3 o = mk(0.f), d = mk(0.f);
    float rx = 0.f, ry = 0.f, rz = 0.f, tmin = 0.f, tmax = 0.f;
    uint32_t cur = 0, slot = 0, lastSeq = 0; int sp = 0;
    uint32_t jobHead = 0, jobCount = 0, enq = 0, done = 0;                       // wave-uniform

    auto popNext = [&]() {                                                       // next stack entry -> cur, or start draining
        if (sp > 0) {
            --sp;
            const int row = min(sp, LS) * 64;
            uint32_t v = stk[row];
            if (sp >= LS) v = ovfBase[(size_t)(sp - LS) * ovfStride + ovfLane];
            cur = v;
        } else state = 2;
    };
    auto push = [&](uint32_t v) {
        if (sp < LS) stk[sp * 64] = v; else ovfBase[(size_t)(sp - LS) * ovfStride + ovfLane] = v;
        ++sp;
    };

    for (;;) {
        // ---- refill idle lanes
        const unsigned long long busy = __ballot(state != 0);
        const int nBusy = __popcll(busy);
        if (nBusy < tune.refill && !exhausted) {
            if (poolNext >= poolEnd) {
                uint32_t b = count;
                if (!staticOnly) {
                    if (lane == 0) b = atomicAdd(head, batch);
                    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                    b = (b < 0xFFFFFFFFu - nWaves * batch) ? b + nWaves * batch : count;
                }
                if (b >= count) { exhausted = true; poolNext = poolEnd = 0; }
                else { poolNext = b; poolEnd = min(b + batch, count); }
            }
            if (!exhausted) {
                const uint32_t take = min((uint32_t)(64 - nBusy), poolEnd - poolNext);
                const unsigned long long idle = ~busy;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                if (state == 0 && rank < take) {
                    const uint32_t qi = poolNext + rank;
                    slot = queue ? queue[qi] : qi;
                    float4 a, b;
                    if (MODE == 0) { a = W.rayA[slot]; b = W.rayB[slot]; }
                    else { const float4 sa = W.shA[slot], sb = W.shB[slot]; a = make_float4(sa.x, sa.y, sa.z, sb.w); b = make_float4(sb.x, sb.y, sb.z, sa.w); }
                    o = mk(a.x, a.y, a.z); d = mk(b.x, b.y, b.z); tmin = a.w; tmax = b.w;
                    if (STATS) cn.rays++;
                    rx = rcpExact(fabsf(d.x) < 1e-20f ? copysignf(1e-20f, d.x) : d.x);
                    ry = rcpExact(fabsf(d.y) < 1e-20f ? copysignf(1e-20f, d.y) : d.y);
                    rz = rcpExact(fabsf(d.z) < 1e-20f ? copysignf(1e-20f, d.z) : d.z);
                    bool go = (root != 0xFFFFFFFFu) && rayIsFinite(o, d);
                    if (go && MODE == 2 && invisibleLightOnSegment(P, T, o, d, rx, ry, rz, tmin, tmax)) {
                        // the reference's walk-through of an invisible light (integrator.cpp:257-278): left to kz_wf_trace<2>
                        literalQueue[atomicAdd(literalCount, 1u)] = slot;
                        go = false;
                    } else if (!go) {
                        // a ray that cannot hit anything (empty scene, non-finite origin / direction)
                        if (MODE == 0) W.hit[slot] = make_float4(KZ_INF, 0.f, 0.f, 0.f);
                        else { const float4 l = W.shL[slot]; W.outR[slot] += l.x; W.outG[slot] += l.y; W.outB[slot] += l.z; }
                    }
                    if (go) {
                        cur = root; sp = 0; state = 1; lastSeq = done;
                        best[lane] = ~0ull;                                       // no hit yet (MODE 2: raised to 0 by an occluding triangle)
                    }
                }
                poolNext += take;
            }
        }
        const unsigned long long trav = __ballot(state == 1);
        if (trav == 0ull && jobCount == 0u && __ballot(state == 2) == 0ull) { if (exhausted) break; continue; }

        // ---- node step for every traversing lane that holds an inner node
        if (state == 1 && !(cur & 0x80000000u)) {
            // the closest hit the job lanes have found for this ray so far (MODE 2: an occluder ends the ray)
            const uint32_t bestHi = (uint32_t)(best[lane] >> 32);
            if (MODE == 2 && bestHi == 0u) { state = 2; sp = 0; }
            else {
                if (MODE == 0) tmax = __uint_as_float(min(__float_as_uint(tmax), bestHi));       // positive floats order as their bits; ~0 = no hit
                if (STATS) cn.nodes++;
                uint32_t key[4]; uint4 refs;
                node4Keys(T, cur, o, rx, ry, rz, tmin, tmax, key, refs);
                bool p0, p1, p2, p3; uint32_t nxt; bool any;
                if (MODE == 2) {                                                     // any hit: children in slot order
                    const bool h0 = key[0] != 0xFFFFFFFFu, h1 = key[1] != 0xFFFFFFFFu, h2 = key[2] != 0xFFFFFFFFu, h3 = key[3] != 0xFFFFFFFFu;
                    any = h0 || h1 || h2 || h3;
                    nxt = h0 ? refs.x : (h1 ? refs.y : (h2 ? refs.z : refs.w));
                    p0 = false; p1 = h1 && h0; p2 = h2 && (h0 || h1); p3 = h3 && (h0 || h1 || h2);
                } else {                                                             // nearest hit child first, the others in slot order
                    const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
                    any = kmin != 0xFFFFFFFFu;
                    nxt = pick4b(refs, kmin);
                    p0 = key[0] != 0xFFFFFFFFu && key[0] != kmin; p1 = key[1] != 0xFFFFFFFFu && key[1] != kmin;
                    p2 = key[2] != 0xFFFFFFFFu && key[2] != kmin; p3 = key[3] != 0xFFFFFFFFu && key[3] != kmin;
                }
                const int c1 = (int)p0, c2 = c1 + (int)p1, c3 = c2 + (int)p2, np = c3 + (int)p3;
                if (sp + 3 <= LS) {                                                  // branch-free: a missed child lands on the scratch row
                    if (MODE != 2) stk[(p0 ? sp : LS) * 64] = refs.x;
                    stk[(p1 ? sp + c1 : LS) * 64] = refs.y; stk[(p2 ? sp + c2 : LS) * 64] = refs.z; stk[(p3 ? sp + c3 : LS) * 64] = refs.w;
                    sp += np;
                } else {
                    if (p0) push(refs.x);
                    if (p1) push(refs.y);
                    if (p2) push(refs.z);
                    if (p3) push(refs.w);
                }
                if (any) cur = nxt; else popNext();
            }
        }
        // ---- room for this step's leaves? run a batch first if the ring could overflow
        bool runJobs = jobCount > KZ_DQ_JOBS - 64;
        if (!runJobs) {
            // ---- lanes that hold a leaf hand it to the job queue and move on
            const bool atLeaf = state == 1 && (cur & 0x80000000u);
            const unsigned long long lm = __ballot(atLeaf);
            if (lm) {
                if (atLeaf) {
                    const uint32_t pos = (jobHead + jobCount + __builtin_amdgcn_mbcnt_hi((uint32_t)(lm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)lm, 0u))) & (KZ_DQ_JOBS - 1);
                    jobLeaf[pos] = cur; jobOwner[pos] = (uint32_t)lane;
                }
                const uint32_t n = (uint32_t)__popcll(lm);
                jobCount += n; enq += n;
                if (atLeaf) { lastSeq = enq; popNext(); }
            }
            // run a batch when a full wave of jobs waits, or when no lane can take a node step
            runJobs = jobCount >= 64u || (jobCount > 0u && __ballot(state == 1 && !(cur & 0x80000000u)) == 0ull);
        }
        if (runJobs) {
            const uint32_t n = min(64u, jobCount);
            const bool mine = (uint32_t)lane < n;
            const uint32_t jp = (jobHead + (uint32_t)lane) & (KZ_DQ_JOBS - 1);
            const uint32_t leaf = mine ? jobLeaf[jp] : 0x80000000u, owner = mine ? jobOwner[jp] : (uint32_t)lane;
            // the owner's ray (every lane takes part in the permutes)
            const int oa = (int)(owner << 2);
            const V3 jo = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(o.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(o.y))),
                             __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(o.z))));
            const V3 jd = mk(__int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(d.x))), __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(d.y))),
                             __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(d.z))));
            const float jtmin = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(tmin)));
            float jtmax = __int_as_float(__builtin_amdgcn_ds_bpermute(oa, __float_as_int(tmax)));
            unsigned long long myKey = ~0ull; uint32_t myTri = 0; float myU = 0.f, myV = 0.f;
            if (mine) {
                const uint32_t bh = (uint32_t)(best[owner] >> 32);
                if (MODE == 0) jtmax = __uint_as_float(min(__float_as_uint(jtmax), bh));
                if (!(MODE == 2 && bh == 0u)) {
                    const uint32_t start = (leaf & 0x7fffffffu) >> 3, cnt = (leaf & 7u) + 1;
                    for (uint32_t i = 0; i < cnt; ++i) {
                        float t, u, v; uint32_t g;
                        if (STATS) cn.tris++;
                        if (!triTest(T.tris + start + i, jo, jd, jtmin, jtmax, t, u, v, g)) continue;
                        if (MODE == 2) { myKey = 0ull; break; }
                        const unsigned long long k = ((unsigned long long)__float_as_uint(t) << 32) | g;
                        if (k < myKey) { myKey = k; myTri = start + i; myU = u; myV = v; jtmax = t; }
                    }
                    if (myKey != ~0ull) atomicMin(&best[owner], myKey);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (MODE == 0 && mine && myKey != ~0ull && best[owner] == myKey) {       // the winner leaves the triangle and its barycentrics
                res[owner] = myTri; res[64 + owner] = __float_as_uint(myU); res[128 + owner] = __float_as_uint(myV);
            }
            jobHead = (jobHead + n) & (KZ_DQ_JOBS - 1); jobCount -= n; done += n;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- publish: stack empty and the last job of the ray has run (sequence numbers wrap consistently)
        if (state == 2 && (int)(done - lastSeq) >= 0) {
            const unsigned long long b = best[lane];
            if (MODE == 0) {
                W.hit[slot] = (b != ~0ull) ? make_float4(__uint_as_float((uint32_t)(b >> 32)), __uint_as_float(res[64 + lane]), __uint_as_float(res[128 + lane]), __uint_as_float((uint32_t)b))        // the key's low word is the gid
                                           : make_float4(KZ_INF, 0.f, 0.f, 0.f);
            } else if (b != 0ull) { const float4 l = W.shL[slot]; W.outR[slot] += l.x; W.outG[slot] += l.y; W.outB[slot] += l.z; }     // nothing on the segment
            state = 0;
        }
    }
    if (STATS) wfStatsFlush(W.stats, cn, 0);
}

// ---- round 4: shade(k) as two kernels (-DKZ_SHADE_SPLIT=1; launched from wfPass, kz_render.hip) ----
// TWO kernels per bounce (round 4; development builds only: -DKZ_EXPERIMENTS -DKZ_SHADE_SPLIT=1): pass A as a kernel of its own at
// 8 waves per SIMD writes the slots of the surviving paths to a dense queue; pass B runs on full waves from that queue, rebuilds the intersection
// record from the hit record and needs neither the 40-KB LDS record stack of the one-kernel form nor its two barriers per round. MEASURED AND REJECTED
// (profiles/r04b_shade_split, same gpurun call, films bit-identical): C4 shade 18.1 -> 20.9 ms per pass (classification 6.1 ms - no faster at 8 or 7 waves
// per SIMD than as pass A of the one kernel, 5.7 - and pass B 14.8 ms against 9.4 + 3.3 of compaction and barriers: the second walk down
// queue -> hit record -> shading record costs more than the record stack did), C3 23.5 -> 28.5 ms. The kernel is bound by the DEPTH of its chains
// of dependent loads, and a split adds two levels.
#ifndef KZ_CLASSIFY_WAVES
#define KZ_CLASSIFY_WAVES 4      // (8 when it was measured, profiles/r04b_shade_split; the double-precision transcendentals of r04m took it to 4: asking for 8 only earned a warning per instantiation)
#endif
template <bool STATS, int EXT>
__global__ __launch_bounds__(KZ_BLOCK) __attribute__((amdgpu_waves_per_eu(KZ_CLASSIFY_WAVES, KZ_CLASSIFY_WAVES))) void kz_wf_classify(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ pixList, uint32_t S,
                                                           uint32_t sampleBegin, int iter, const uint32_t *__restrict__ queue,
                                                           const uint32_t *__restrict__ countPtr, uint32_t countImm,
                                                           uint32_t *__restrict__ survQueue, uint32_t *__restrict__ survCount) {
    __shared__ uint32_t s_buf[KZ_WF_QCAP]; __shared__ uint32_t s_n, s_gb;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    WfAppender ap = {s_buf, &s_n, &s_gb, survQueue, survCount};
    const uint32_t count = countPtr ? *countPtr : countImm;
    const bool compact = !EXT && !P.regularization;
    Counters cn = {0, 0, 0, 0, 0, 0};
    KzSst sst;
    int round = 0;
    for (uint32_t base = blockIdx.x * KZ_BLOCK; base < count; base += gridDim.x * KZ_BLOCK) {
        bool survivor = false;
        uint32_t slot = 0;
        if (base + threadIdx.x < count) {
            const uint32_t qi = base + threadIdx.x;
            slot = queue ? queue[qi] : qi;
            Its its;
            { uint32_t modelKey_ = 0; survivor = wfClassify<STATS, EXT>(P, T, W, pixList, S, sampleBegin, iter, compact, slot, its, modelKey_, cn, sst); }
        }
        sst.mark(0);
        ap.push(survivor, slot);
        // The staging buffer takes KZ_WF_ROUNDS rounds of 256 entries: it is flushed every so many rounds whatever it holds, so the decision needs
        // no look at the shared count (a wave that has run ahead into the next round may already be adding to it) and the rounds between two
        // flushes need no barrier at all.
        if (++round == KZ_WF_ROUNDS) {
            round = 0;
            __syncthreads();
            const uint32_t n = s_n;                         // (stable: every wave's next push is behind the barriers below)
            if (threadIdx.x == 0) s_gb = atomicAdd(survCount, n);
            __syncthreads();
            const uint32_t gb = s_gb;
            for (uint32_t i = threadIdx.x; i < n; i += KZ_BLOCK) survQueue[gb + i] = s_buf[i];
            __syncthreads();
            if (threadIdx.x == 0) s_n = 0;
            __syncthreads();
        }
        sst.mark(1);
    }
    ap.maybeFlush(true);
    if (STATS) wfStatsFlush(W.stats, cn, 0);
    sst.flush(W.stats);
}

template <bool STATS, int EXT>
__global__ __launch_bounds__(KZ_BLOCK, (EXT ? 3 : KZ_SHADE_WAVES)) void kz_wf_shade_b(KzParams P, KzDevTables T, KzWf W, const uint32_t *__restrict__ pixList, uint32_t S,
                                                        uint32_t sampleBegin, int iter, const uint32_t *__restrict__ survQueue, const uint32_t *__restrict__ survCount,
                                                        uint32_t *__restrict__ nextQueue, uint32_t *__restrict__ nextCount,
                                                        uint32_t *__restrict__ shadowQueue, uint32_t *__restrict__ shadowCount) {
    __shared__ uint32_t s_bufN[KZ_WF_QCAP], s_bufS[KZ_WF_QCAP]; __shared__ uint32_t s_nN, s_nS, s_gbN, s_gbS;
    if (threadIdx.x == 0) { s_nN = 0; s_nS = 0; }
    __syncthreads();
    WfAppender apN = {s_bufN, &s_nN, &s_gbN, nextQueue, nextCount}, apS = {s_bufS, &s_nS, &s_gbS, shadowQueue, shadowCount};
    WfQueuePair qp = {s_bufN, s_bufS, &s_nN, &s_nS, &s_gbN, &s_gbS, nextQueue, nextCount, shadowQueue, shadowCount};
    const uint32_t count = *survCount;
    const bool compact = !EXT && !P.regularization;
    Counters cn = {0, 0, 0, 0, 0, 0};
    KzSst sst;
    int round = 0;
    for (uint32_t base = blockIdx.x * KZ_BLOCK; base < count; base += gridDim.x * KZ_BLOCK) {
        bool pushNext = false, pushShadow = false;
        uint32_t slot = 0;
        if (base + threadIdx.x < count) {
            slot = kzLoadStream(survQueue + base + threadIdx.x);
            const float4 h = kzLoadStream(&W.hit[slot]);
            RawHit rh; rh.t = h.x; rh.u = h.y; rh.v = h.z; rh.tri = 0; rh.gid = __float_as_uint(h.w);
            Its its; postIntersect<false>(T, rh, its);                 // (counted by kz_wf_classify)
            sst.markw(12);
            wfShadeSurvivor<STATS, EXT>(P, T, W, pixList, S, sampleBegin, iter, compact, slot, its, pushNext, pushShadow, cn, sst);
        }
        sst.mark(7);
        apN.push(pushNext, slot); apS.push(pushShadow, slot);
        if (++round == KZ_WF_ROUNDS) {                     // (see kz_wf_classify)
            round = 0;
            __syncthreads();
            qp.flushAll();
        }
        sst.mark(8);
    }
    __syncthreads();
    qp.flush(true);
    if (STATS) wfStatsFlush(W.stats, cn, 0);
    sst.flush(W.stats);
}

// ---- round 4: sorted traversal queues (needs -DKZ_EXPERIMENTS -DKZ_SORT_EXPERIMENT; the call sites are in wfPass, kz_render.hip) ----
#ifdef KZ_SORT_EXPERIMENT
// Development build only (-DKZ_SORT_EXPERIMENT; profiles/r04d_sorted_queues): the UPPER BOUND of what ray reordering can buy the per-lane traversal kernels
// ("dual queues"). After shade(k) the bounce-ray queue and the shadow-ray queue are each sorted by (Morton code of the ray origin in the
// scene box, direction octant) with hipcub's radix sort into SEPARATE queues that only the traversal launches read - the next shade keeps the slot-order
// queue. The count is fetched with a host sync: this measures the traversal on sorted input, not a pipeline one would ship.
#include <hipcub/hipcub.hpp>
__device__ __forceinline__ uint32_t kzPart1By2(uint32_t x) { x &= 0x3ffu; x = (x | (x << 16)) & 0x30000ffu; x = (x | (x << 8)) & 0x300f00fu; x = (x | (x << 4)) & 0x30c30c3u; x = (x | (x << 2)) & 0x9249249u; return x; }
__global__ void kz_sort_keys(KzWf W, int shadow, const uint32_t *__restrict__ q, uint32_t n, float lx, float ly, float lz, float sx, float sy, float sz, int bits, int useDir, uint32_t *__restrict__ keys) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t slot = q[i];
    const float4 a = shadow ? W.shA[slot] : W.rayA[slot], b = shadow ? W.shB[slot] : W.rayB[slot];
    const uint32_t qx = (uint32_t)fminf(fmaxf((a.x - lx) * sx, 0.f), 1023.f), qy = (uint32_t)fminf(fmaxf((a.y - ly) * sy, 0.f), 1023.f), qz = (uint32_t)fminf(fmaxf((a.z - lz) * sz, 0.f), 1023.f);
    uint32_t m = (kzPart1By2(qx) | (kzPart1By2(qy) << 1) | (kzPart1By2(qz) << 2)) >> (30 - 3 * bits);
    const uint32_t oct = (b.x < 0.f ? 1u : 0u) | (b.y < 0.f ? 2u : 0u) | (b.z < 0.f ? 4u : 0u);
    keys[i] = useDir == 2 ? ((oct << (3 * bits)) | m) : useDir == 1 ? ((m << 3) | oct) : m;
}
struct KzSortExp {
    uint32_t *keysIn = nullptr, *keysOut = nullptr, *qOut[2] = {nullptr, nullptr}; void *temp = nullptr; size_t tempBytes = 0, cap = 0;
    float lo[3], sc[3]; int bits = 6, useDir = 1; bool on = true; double sortMs = 0; hipEvent_t e0 = nullptr, e1 = nullptr;
    int ensure(KzScene *scene, size_t need) {
        if (const char *e = std::getenv("KZ_SORT_BITS")) bits = std::max(1, std::min(10, atoi(e)));
        if (const char *e = std::getenv("KZ_SORT_DIR")) useDir = atoi(e);
        if (const char *e = std::getenv("KZ_SORT_ON")) on = atoi(e) != 0;
        if (need <= cap) return KZ_OK;
        const KzNode &r = scene->nodes[scene->prm.rootRef & 0x7fffffffu];
        const float blo[3] = {std::min(r.q[0], r.q[6]), std::min(r.q[1], r.q[7]), std::min(r.q[2], r.q[8])}, bhi[3] = {std::max(r.q[3], r.q[9]), std::max(r.q[4], r.q[10]), std::max(r.q[5], r.q[11])};
        for (int a = 0; a < 3; ++a) { lo[a] = blo[a]; sc[a] = 1024.0f / std::max(1e-20f, bhi[a] - blo[a]); }
        HIP_TRY(hipDeviceSynchronize());
        for (void *p_ : {(void *)keysIn, (void *)keysOut, (void *)qOut[0], (void *)qOut[1], temp}) if (p_) (void)hipFree(p_);
        KZ_ALLOC(&keysIn, need * 4); KZ_ALLOC(&keysOut, need * 4); KZ_ALLOC(&qOut[0], need * 4); KZ_ALLOC(&qOut[1], need * 4);
        tempBytes = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, keysIn, keysOut, qOut[0], qOut[1], (int)need, 0, 32, (hipStream_t)0);
        KZ_ALLOC(&temp, tempBytes);
        cap = need;
        if (!e0) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); }
        return KZ_OK;
    }
    // sorted copy of queue q (count at *countPtr on the device) -> qOut[which]; returns the host copy of the count
    int sort(hipStream_t stream, const KzWf &W, int shadow, const uint32_t *q, const uint32_t *countPtr, int which, uint32_t *nOut) {
        uint32_t n = 0;
        HIP_TRY(hipMemcpyAsync(&n, countPtr, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        *nOut = n;
        if (!n) return KZ_OK;
        HIP_TRY(hipEventRecord(e0, stream));
        hipLaunchKernelGGL(kz_sort_keys, dim3((n + 255) / 256), dim3(256), 0, stream, W, shadow, q, n, lo[0], lo[1], lo[2], sc[0], sc[1], sc[2], bits, useDir, keysIn);
        size_t tb = tempBytes;
        HIP_TRY(hipcub::DeviceRadixSort::SortPairs(temp, tb, keysIn, keysOut, q, qOut[which], (int)n, 0, 3 * bits + (useDir ? 3 : 0), stream));
        HIP_TRY(hipEventRecord(e1, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, e0, e1)); sortMs += ms;
        return KZ_OK;
    }
};
static KzSortExp g_sortExp;
#endif
