// kz_device.hip — hand-written HIP kernels (gfx950 / CDNA4) and their launch code for the path_mis hot path:
//   primary-ray generation -> BVH2 traversal with Moeller-Trumbore leaf tests -> post-intersection ->
//   kiss/diffuse BSDF eval+sample -> MIS NEE with the invisible-light walk-through -> Russian roulette
//   -> per-sample radiance -> deterministic film reconstruction (ImageBlock::put semantics).
// Reference region replaced: src/kazen/renderer.cpp:85-133 and everything it calls.
//
// Execution model (wave64): one lane = one (pixel, sample) path; items are ordered pixel-major so the lanes
// of a wave share a pixel neighbourhood (coherent primary rays, shared top-of-tree node packets in L1/L2).
// Per-lane traversal stacks live in LDS ([depth][lane] -> conflict-free columns). No MFMA: this is branchy
// pointer chasing bound by HBM/L2 latency and bandwidth, not a contraction.
#include <hip/hip_runtime.h>

#include "kz_internal.h"
#include "kz_devfn.h"
#include "kz_wavefront.h"
#ifdef KZ_EXPERIMENTS
#include "kz_experiments.h"
#endif

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


// a1/a2 renderBlock + renderSample (renderer.cpp:20-69): item = pixLinear * S + sampleOffset
template <bool STATS, bool EXT>
__global__ __launch_bounds__(KZ_BLOCK) void kz_path_megakernel(KzParams P, KzDevTables T, const uint32_t *__restrict__ pixList,
                                                               uint32_t nItems, uint32_t S, uint32_t sampleBegin, const uint32_t *__restrict__ itemSample,
                                                               float *__restrict__ outJx, float *__restrict__ outJy, float *__restrict__ outR,
                                                               float *__restrict__ outG, float *__restrict__ outB,
                                                               unsigned long long *__restrict__ stats) {
    __shared__ uint32_t s_stack[KZ_STACK_DEPTH * KZ_BLOCK];
    const uint32_t item = blockIdx.x * KZ_BLOCK + threadIdx.x;
    Counters cn = {0, 0, 0, 0, 0, 0};
    if (item < nItems) {
        const uint32_t pl = item / S, so = item - pl * S;
        const uint32_t pxy = pixList[pl];
        const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
        Sampler smp; smp.type = P.samplerType;
        smp.generateSample(P, T, px, py, itemSample ? itemSample[item] : sampleBegin + so);
        float jx, jy; smp.nextPixel2D(P, T, jx, jy);
        const float sx = (float)px + jx, sy = (float)py + jy;
        float ax, ay; smp.next2D(P, T, ax, ay);                  // aperture sample, always consumed (renderer.cpp:28)
        V3 ro, rd; float mint, maxt;
        cameraRay(P, sx, sy, ax, ay, ro, rd, mint, maxt);
        V3 L = pathLi<STATS, EXT>(P, T, smp, ro, rd, mint, maxt, s_stack + threadIdx.x, cn);
        outJx[item] = jx; outJy[item] = jy; outR[item] = L.x; outG[item] = L.y; outB[item] = L.z;
        if (STATS) {
            bool valid = L.x >= 0 && L.y >= 0 && L.z >= 0 && isfinite(L.x) && isfinite(L.y) && isfinite(L.z);
            if (!valid) cn.dropped++;
        }
    }
    if (STATS) {
        // wave reduction then one atomic per counter per wave
        unsigned long long v[7] = {item < nItems ? 1ull : 0ull, cn.rays, cn.nodes, cn.tris, cn.hits, cn.lsamples, cn.dropped};
        for (int k = 0; k < 7; ++k) {
            unsigned long long x = v[k];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(&stats[k], x);
        }
    }
}

// ============================================================================================
// a25 ImageBlock::put as a deterministic gather (block.cpp:56-85). One workgroup = 16x16 film pixels. The
// samples of the (16+taps-1)^2 source pixels that can reach them are staged through LDS in chunks (coalesced
// global reads, each sample record read once per workgroup instead of once per film pixel), then every thread
// sums, in a fixed order, the samples whose filter footprint covers its pixel. Positions are formed
// block-relative exactly as the reference does (32x32 blocks at multiples of KAZEN_BLOCK_SIZE), so the weights
// are bit-identical to ImageBlock::put; only the order of the float additions differs (H10).
// ============================================================================================
#define KZ_FILM_TILE 16
#define KZ_FILM_RMAX (KZ_FILM_TILE + KZ_MAX_FILTER_TAPS - 1)
// The filter weight of a sample is separable and, per axis, depends only on the sample and on WHICH of its `taps` neighbour
// columns (rows) the film pixel is: the staging pass evaluates validity (Color3f::isValid), the bounds test and the table
// look-up of block.cpp:64-80 once per (sample, tap) — 2*taps evaluations per sample instead of 2*taps^2 — and the gather pass
// is left with five LDS reads and the multiply-adds of block.cpp:84. A sample that is invalid, absent or out of bounds carries
// weight 0 and adds an exact zero, so the sums are the ones the reference forms.
__global__ __launch_bounds__(256) void kz_film_gather(KzParams P, const float *__restrict__ filter, const int32_t *__restrict__ pixIndex,
                                                      uint32_t p0, uint32_t nPixPass, uint32_t S, int chunk, const float *__restrict__ inJx, const float *__restrict__ inJy,
                                                      const float *__restrict__ inR, const float *__restrict__ inG, const float *__restrict__ inB,
                                                      float4 *__restrict__ film) {
    __shared__ float s_filter[KZ_FILTER_RESOLUTION + 1];
    __shared__ int32_t s_pl[KZ_FILM_RMAX * KZ_FILM_RMAX];
    extern __shared__ float s_samp[];              // [3 + 2*taps][chunk][R*R]: r g b | wx[taps] | wy[taps]
    const int tid = threadIdx.x;
    if (tid <= KZ_FILTER_RESOLUTION) s_filter[tid] = filter[tid];
    const int cols = P.width + 2 * P.border, rows = P.height + 2 * P.border;
    const int taps = P.tapHi - P.tapLo + 1;
    const int R = KZ_FILM_TILE + taps - 1, RR = R * R;
    // film tile origin (film coordinates) and the image-space origin of the source region
    const int fx0 = blockIdx.x * KZ_FILM_TILE, fy0 = blockIdx.y * KZ_FILM_TILE;
    const int sx0 = fx0 - P.border + P.tapLo, sy0 = fy0 - P.border + P.tapLo;
    bool anySrc = false;
    for (int q = tid; q < RR; q += 256) {
        const int x = sx0 + q % R, y = sy0 + q / R;
        int pl = -1;
        if (x >= 0 && x < P.width && y >= 0 && y < P.height) pl = pixIndex[y * P.width + x];
        if (pl >= 0) { pl -= (int)p0; if (pl < 0 || pl >= (int)nPixPass) pl = -1; }       // a pass covers pixels [p0, p0 + nPixPass) of the pixel list
        s_pl[q] = pl;
        anySrc |= pl >= 0;
    }
    if (!__syncthreads_or(anySrc)) return;        // nothing of this pass can reach the tile
    const int lx = tid & 15, ly = tid >> 4;
    const int fx = fx0 + lx, fy = fy0 + ly;
    const bool inFilm = fx < cols && fy < rows;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const float r = P.filterRadius, lf = P.lookupFactor;
    const int plane = chunk * RR;
    float *s_wx = s_samp + 3 * plane, *s_wy = s_samp + (3 + taps) * plane;
    for (uint32_t sBase = 0; sBase < S; sBase += (uint32_t)chunk) {
        const int ch = (int)min((uint32_t)chunk, S - sBase);
        __syncthreads();
        for (int i = tid; i < RR * ch; i += 256) {
            const int q = i / ch, s = i - q * ch;
            const int pl = s_pl[q];
            if (pl < 0) continue;                                            // the gather pass skips these source pixels
            const size_t gi = (size_t)pl * S + sBase + s;
            const float jx = inJx[gi], jy = inJy[gi], cr = inR[gi], cg = inG[gi], cb = inB[gi];
            const bool valid = cr >= 0.f && cg >= 0.f && cb >= 0.f && isfinite(cr) && isfinite(cg) && isfinite(cb);   // Color3f::isValid
            const int o = s * RR + q;
            s_samp[o] = valid ? cr : 0.f; s_samp[plane + o] = valid ? cg : 0.f; s_samp[2 * plane + o] = valid ? cb : 0.f;
            const int px = sx0 + q % R, py = sy0 + q / R;
            const int bx0 = px & ~31, by0 = py & ~31;                        // the reference block this sample is rendered in
            const float posx = ((float)px + jx) - 0.5f - (float)(bx0 - P.border), posy = ((float)py + jy) - 0.5f - (float)(by0 - P.border);   // block.cpp:64-67
            const float lox = ceilf(posx - r), hix = floorf(posx + r), loy = ceilf(posy - r), hiy = floorf(posy + r);                     // block.cpp:70-73
            for (int t = 0; t < taps; ++t) {
                // the film pixel that sees this source pixel through tap t: f = p + border - tapLo - t
                const float xb = (float)(px + P.border - P.tapLo - t - bx0), yb = (float)(py + P.border - P.tapLo - t - by0);
                float wx = 0.f, wy = 0.f;
                if (valid && !(xb < lox || xb > hix)) wx = s_filter[(int)(fabsf(xb - posx) * lf)];                                       // block.cpp:77-80
                if (valid && !(yb < loy || yb > hiy)) wy = s_filter[(int)(fabsf(yb - posy) * lf)];
                s_wx[t * plane + o] = wx; s_wy[t * plane + o] = wy;
            }
        }
        __syncthreads();
        if (!inFilm) continue;
        for (int dy = 0; dy < taps; ++dy) {
            for (int dx = 0; dx < taps; ++dx) {
                const int q = (ly + dy) * R + (lx + dx);
                if (s_pl[q] < 0) continue;
                for (int s = 0; s < ch; ++s) {
                    const int o = s * RR + q;
                    const float cr = s_samp[o], cg = s_samp[plane + o], cb = s_samp[2 * plane + o];
                    const float wx = s_wx[dx * plane + o], wy = s_wy[dy * plane + o];
                    acc.x += cr * wx * wy; acc.y += cg * wx * wy; acc.z += cb * wx * wy; acc.w += 1.0f * wx * wy;   // block.cpp:84
                }
            }
        }
    }
    if (inFilm) {
        float4 *dst = film + (size_t)fy * cols + fx;
        float4 o = *dst;
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *dst = o;
    }
}

// ---- a25 in two kernels for filters of at most 5 taps per axis (every default of the reference: gaussian / mitchell radius 2, tent, box) ----
// kz_film_gather stages the samples of a 20x20 pixel neighbourhood three at a time (LDS) and so reads 12-byte pieces of the 256-byte
// sample rows: rocprofv3 shows 23.5 GB fetched per pass for 2.65 GB of records, an HBM-bound 6.2 ms (profiles/r02b_packet_primary).
// Here every record is read exactly once:
//   kz_film_taps   one THREAD per SOURCE pixel, one wave per 64 consecutive pixels of the pass's pixel list (an 8x8 block). The wave copies
//                  8 samples of its 64 pixels at a time into LDS with coalesced 32-byte pieces, transposed to [sample][pixel]; each thread
//                  then walks ITS pixel's samples in sample order - validity, the separable filter weights of block.cpp:64-80 per tap, the
//                  taps x taps weighted products of block.cpp:84 - into taps^2 (rgb*w, w) accumulators that never leave its registers,
//                  and stores them tap-major ([tap][pixel]: coalesced).
//   kz_film_apply  one thread per FILM pixel: adds, in a fixed (row, column) tap order, the tap sums of the <= taps^2 source pixels that reach it.
// Deterministic (per pixel the samples are added in index order, as the reference's put() loop does; fixed tap order in the second
// kernel); the weights are the ones kz_film_gather forms (block-relative positions, same table look-ups).
#define KZ_TAPS_MAX 5                        // gaussian / mitchell radius 2: taps -2..2
#ifndef KZ_TAPS_CHUNK
#define KZ_TAPS_CHUNK 8                      // samples per staging round (32 B of every sample row). The staging area is what limits the waves per CU: 16 (20.8 KB
#endif                                       // per wave, 7 per CU) 2.15 ms per pass, 8 (10.4 KB, 15 per CU) 1.95, 4 (16-B pieces) 3.46 - same call, C4 and C3 alike
template <int TAPS>
__global__ __launch_bounds__(64) void kz_film_taps(KzParams P, const float *__restrict__ filter, const uint32_t *__restrict__ pixList, uint32_t nPix, uint32_t S,
                                                   const float *__restrict__ inJx, const float *__restrict__ inJy, const float *__restrict__ inR,
                                                   const float *__restrict__ inG, const float *__restrict__ inB, float4 *__restrict__ tapSums) {
    __shared__ float s_filter[KZ_FILTER_RESOLUTION + 1];
    __shared__ float s_in[5][KZ_TAPS_CHUNK][65];                       // [array][sample][pixel], rows padded against bank conflicts of the transposing store
    const int lane = threadIdx.x;
    if (lane <= KZ_FILTER_RESOLUTION) s_filter[lane] = filter[lane];
    const uint32_t pl0 = blockIdx.x * 64u, pl = pl0 + (uint32_t)lane;
    const bool havePixel = pl < nPix;
    const uint32_t pxy = havePixel ? pixList[pl] : 0u;
    const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
    const int bx0 = px & ~31, by0 = py & ~31;                          // the reference block this pixel is rendered in
    const float r = P.filterRadius, lf = P.lookupFactor;
    float xb[TAPS], yb[TAPS];                                          // block-relative film coordinates this pixel reaches through tap t
#pragma unroll
    for (int t = 0; t < TAPS; ++t) { xb[t] = (float)(px + P.border - P.tapLo - t - bx0); yb[t] = (float)(py + P.border - P.tapLo - t - by0); }
    float4 acc[TAPS * TAPS];
#pragma unroll
    for (int i = 0; i < TAPS * TAPS; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *const in[5] = {inJx, inJy, inR, inG, inB};
    const uint32_t nRows = min(64u, nPix - min(nPix, pl0));            // pixels of this wave
    for (uint32_t c0 = 0; c0 < S; c0 += KZ_TAPS_CHUNK) {
        const uint32_t n = min((uint32_t)KZ_TAPS_CHUNK, S - c0);
        __syncthreads();
        for (uint32_t i = lane; i < nRows * KZ_TAPS_CHUNK; i += 64u) {  // consecutive lanes: consecutive samples of one pixel (32-B pieces), then the next pixel
            const uint32_t p = i / KZ_TAPS_CHUNK, k = i % KZ_TAPS_CHUNK;
            if (k < n) {
                const size_t gi = (size_t)(pl0 + p) * S + c0 + k;
#pragma unroll
                for (int a = 0; a < 5; ++a) s_in[a][k][p] = in[a][gi];
            }
        }
        __syncthreads();
        if (havePixel) {
            for (uint32_t k = 0; k < n; ++k) {
                const float jx = s_in[0][k][lane], jy = s_in[1][k][lane];
                float cr = s_in[2][k][lane], cg = s_in[3][k][lane], cb = s_in[4][k][lane];
                const bool valid = cr >= 0.f && cg >= 0.f && cb >= 0.f && isfinite(cr) && isfinite(cg) && isfinite(cb);   // Color3f::isValid
                if (!valid) continue;                                  // an invalid sample carries weight 0 everywhere: adds exact zeros
                const float posx = ((float)px + jx) - 0.5f - (float)(bx0 - P.border), posy = ((float)py + jy) - 0.5f - (float)(by0 - P.border);   // block.cpp:64-67
                const float lox = ceilf(posx - r), hix = floorf(posx + r), loy = ceilf(posy - r), hiy = floorf(posy + r);                     // block.cpp:70-73
                float wx[TAPS], wy[TAPS];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    wx[t] = !(xb[t] < lox || xb[t] > hix) ? s_filter[(int)(fabsf(xb[t] - posx) * lf)] : 0.f;                                 // block.cpp:77-80
                    wy[t] = !(yb[t] < loy || yb[t] > hiy) ? s_filter[(int)(fabsf(yb[t] - posy) * lf)] : 0.f;
                }
#pragma unroll
                for (int ty = 0; ty < TAPS; ++ty)
#pragma unroll
                    for (int tx = 0; tx < TAPS; ++tx) {
                        float4 &a = acc[ty * TAPS + tx];
                        a.x += cr * wx[tx] * wy[ty]; a.y += cg * wx[tx] * wy[ty]; a.z += cb * wx[tx] * wy[ty]; a.w += 1.0f * wx[tx] * wy[ty];     // block.cpp:84
                    }
            }
        }
    }
    if (havePixel) {
#pragma unroll
        for (int i = 0; i < TAPS * TAPS; ++i) tapSums[(size_t)i * nPix + pl] = acc[i];
    }
}

__global__ __launch_bounds__(256) void kz_film_apply(KzParams P, const int32_t *__restrict__ pixIndex, const float4 *__restrict__ tapSums, uint32_t p0, uint32_t nPix, float4 *__restrict__ film) {
    const int cols = P.width + 2 * P.border, rows = P.height + 2 * P.border;
    const int fx = blockIdx.x * 16 + (threadIdx.x & 15), fy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (fx >= cols || fy >= rows) return;
    const int taps = P.tapHi - P.tapLo + 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    bool any = false;
    for (int ty = 0; ty < taps; ++ty) {
        const int y = fy - P.border + P.tapLo + ty;                    // the source pixel that reaches this film pixel through tap (tx, ty)
        if (y < 0 || y >= P.height) continue;
        for (int tx = 0; tx < taps; ++tx) {
            const int x = fx - P.border + P.tapLo + tx;
            if (x < 0 || x >= P.width) continue;
            const int32_t pl = pixIndex[y * P.width + x] - (int32_t)p0;      // the pass covers pixels [p0, p0 + nPix) of the pixel list
            if (pl < 0 || pl >= (int32_t)nPix) continue;
            const float4 t = tapSums[(size_t)(ty * taps + tx) * nPix + (size_t)pl];
            acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            any = true;
        }
    }
    if (any) {
        float4 *dst = film + (size_t)fy * cols + fx;
        float4 o = *dst;
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
        *dst = o;
    }
}

// The film rects of a tile list, packed: tile t contributes its (h + 2b) x (w + 2b) rect (the tile with its filter apron) as consecutive rows
// at offsets[t] (in float4s). The aprons of neighbouring tiles of the list overlap in the film; a texel that an EARLIER tile of the list has
// already carried is written as zero, so that the sum of the packed rects is exactly the film over the union of the rects (each texel
// once). `prev` lists, per tile, the earlier tiles whose rect overlaps it. One workgroup per (tile, row).
struct KzTileRect { int32_t x0, y0, w, h; uint32_t offset; uint32_t prevStart, prevCount; };
__global__ __launch_bounds__(128) void kz_film_pack(const float4 *__restrict__ film, int cols, const KzTileRect *__restrict__ rects, const uint32_t *__restrict__ prev,
                                                    int border, float4 *__restrict__ out) {
    const KzTileRect r = rects[blockIdx.y];
    const int rw = r.w + 2 * border, rh = r.h + 2 * border;
    const int row = blockIdx.x;
    if (row >= rh) return;
    const int fy = r.y0 + row;
    const float4 *src = film + (size_t)fy * cols + r.x0;
    float4 *dst = out + r.offset + (size_t)row * rw;
    for (int x = threadIdx.x; x < rw; x += blockDim.x) {
        const int fx = r.x0 + x;
        bool mine = true;
        for (uint32_t k = 0; k < r.prevCount; ++k) {
            const KzTileRect q = rects[prev[r.prevStart + k]];
            if (fx >= q.x0 && fx < q.x0 + q.w + 2 * border && fy >= q.y0 && fy < q.y0 + q.h + 2 * border) { mine = false; break; }
        }
        dst[x] = mine ? src[x] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

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


// Film -> 8-bit sRGB raster: Color4f::divideByFilterWeight (color.h:94-99), Color3f::toSRGB (common.cpp:351-366) and the
// clamp + truncation of Bitmap::savePNG (bitmap.cpp:45-52). One pixel per lane, coalesced float4 reads.
__global__ void kz_film_srgb8(const float4 *__restrict__ film, int width, int height, int border, uint8_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint32_t)(width * height)) return;
    const int y = (int)(i / (uint32_t)width), x = (int)(i - (uint32_t)y * (uint32_t)width);
    const float4 px = film[(size_t)(y + border) * (size_t)(width + 2 * border) + (size_t)(x + border)];
    float c[3] = {0.f, 0.f, 0.f};
    if (px.w != 0.f) { c[0] = px.x / px.w; c[1] = px.y / px.w; c[2] = px.z / px.w; }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = c[k];
        const float t = v <= 0.0031308f ? 12.92f * v : (1.0f + 0.055f) * powf(v, 1.0f / 2.4f) - 0.055f;
        const float s = 255.f * t;
        out[3 * (size_t)i + k] = (uint8_t)(s < 0.f ? 0.f : (s > 255.f ? 255.f : s));
    }
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
    NMap nm; surfaceSetup<true>(T, its, m, nm);
    const V3 a = mk(wi[3 * i], wi[3 * i + 1], wi[3 * i + 2]), b = mk(wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]);
    V3 e = surfEval<true>(m, nm, its, a, b, acc[i]);
    evalOut[3 * i] = e.x; evalOut[3 * i + 1] = e.y; evalOut[3 * i + 2] = e.z;
    pdfOut[i] = surfPdf<true>(m, nm, its, a, b, acc[i], true);
    V3 d; bool alive, discrete, solid; float etaScale, pdfS;
    V3 w = surfSample<true>(m, nm, its, a, acc[i], s3[3 * i], s3[3 * i + 1], s3[3 * i + 2], d, alive, discrete, etaScale, pdfS, solid);
    const bool zero = w.x == 0.f && w.y == 0.f && w.z == 0.f;
    float *o = sampleOut + 8 * i;
    o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = d.x; o[4] = d.y; o[5] = d.z; o[6] = alive ? 1.f : 0.f;
    o[7] = (!alive || zero) ? 0.f : (pdfS >= 0.f ? pdfS : surfPdf<true>(m, nm, its, a, d, acc[i], solid));      // integrator.cpp:314
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

// ============================================================================================
// host side: replicas (one device state per GPU the scene is resident on), upload, passes, the multi-device driver
// ============================================================================================
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return kz_fail(KZ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

// Every device allocation of the library goes through here (kz_debug_fail_alloc can make the nth one fail).
static thread_local int g_failAlloc = 0;
static hipError_t kzMalloc(void **p, size_t bytes) {
    *p = nullptr;
    if (g_failAlloc > 0 && --g_failAlloc == 0) return hipErrorOutOfMemory;
    return hipMalloc(p, bytes);
}
#define KZ_ALLOC(pp, bytes) do { hipError_t e_ = kzMalloc((void **)(pp), (bytes)); if (e_ != hipSuccess) \
    return kz_fail(e_ == hipErrorOutOfMemory ? KZ_ERR_OOM : KZ_ERR_HIP, "device allocation of %zu bytes failed: %s", (size_t)(bytes), hipGetErrorString(e_)); } while (0)
// A device buffer that is released on every way out of the call that made it.
struct DevMem {
    void *p = nullptr;
    DevMem() = default;
    DevMem(const DevMem &) = delete; DevMem &operator=(const DevMem &) = delete;
    ~DevMem() { if (p) (void)hipFree(p); }
    template <class Tp> Tp *as() const { return (Tp *)p; }
};

struct EventPair { hipEvent_t a, b; };
// path state + sample records + stage events of one pass in flight
struct PassCtx {
    KzWf wf{}; std::vector<void *> wfAllocs; size_t wfCap = 0;
    float *samp = nullptr; size_t sampCap = 0;                   // five SoA planes: jx | jy | r | g | b
    float *taps = nullptr; size_t tapsCap = 0;                   // kz_film_taps: taps^2 float4 per pixel of the tile set
    uint32_t *litQueue = nullptr; size_t litCap = 0;             // kz_wf_trace_dq<2>: shadow rays that need the literal walk-through
    uint32_t *ovf = nullptr; size_t ovfCap = 0;
    uint2 *beamEntries = nullptr, *beamCount = nullptr; size_t beamCap = 0;      // kz_wf_beam: leaf lists of the pixels of a chunk ...
    uint64_t beamGen = 0; uint32_t beamP0 = 0, beamN = 0;                                  // ... and the chunk (tile-set generation, first pixel, pixels) they were built for
    uint64_t sharedSeen = 0;                                                              // the generation of the replica's shared lists this context's stream has waited for
    std::vector<hipEvent_t> stageEv; std::vector<int> stageKind; size_t stageUsed = 0;
    size_t bytes() const { return wfCap * (8 * sizeof(float4) + sizeof(uint4) + 3 * sizeof(uint32_t)) + sampCap * 5 * sizeof(float) + ovfCap * sizeof(uint32_t) + tapsCap * 400 + litCap * 4 + beamCap * (KZ_BEAM_CAP + 1) * sizeof(uint2); }
    void release() {
        for (void *p : wfAllocs) (void)hipFree(p);
        wfAllocs.clear(); wfCap = 0; wf = KzWf{};
        if (samp) (void)hipFree(samp); samp = nullptr; sampCap = 0;
        if (ovf) (void)hipFree(ovf); ovf = nullptr; ovfCap = 0;
        if (taps) (void)hipFree(taps); taps = nullptr; tapsCap = 0;
        if (litQueue) (void)hipFree(litQueue); litQueue = nullptr; litCap = 0;
        if (beamEntries) (void)hipFree(beamEntries); beamEntries = nullptr; if (beamCount) (void)hipFree(beamCount); beamCount = nullptr; beamCap = 0; beamGen = 0;
    }
};
struct KzDeviceState {
    int device = -1;
    KzDevTables T{};
    std::vector<void *> allocs;
    float4 *film = nullptr; size_t filmPixels = 0;
    uint8_t *srgb = nullptr;                                     // staging raster of kz_film_to_srgb8 (allocated on first use)
    float4 *packDev = nullptr; size_t packCap = 0; KzTileRect *rectsDev = nullptr; size_t rectsCap = 0; uint32_t *prevDev = nullptr; size_t prevCap = 0;      // kz_film_download_tiles: packed tile rects + their tables
    float4 *packHost = nullptr; size_t packHostCap = 0;           // pinned staging of the same (D2H at link rate)
    uint32_t *pixList = nullptr; int32_t *pixIndex = nullptr; size_t pixCap = 0; uint32_t nPix = 0;
    std::vector<KzTile> curTiles; bool tilesValid = false; uint64_t tileGen = 0;      // tileGen: bumped whenever the pixel list changes
    unsigned long long *stats = nullptr; bool statsOn = false;
    hipStream_t lastStream = nullptr;
    int numCU = 256; size_t totalMem = 0;
    PassCtx ctx[KZ_MAX_PASSES_IN_FLIGHT];
    std::vector<EventPair> events; size_t eventsUsed = 0;
    hipStream_t passStream[KZ_MAX_PASSES_IN_FLIGHT] = {}; hipEvent_t evFork = nullptr, evFilm[KZ_MAX_PASSES_IN_FLIGHT] = {}, evCallA = nullptr, evCallB = nullptr;
    int lastCtx = 0; bool lastDual = false; int streamMode = 0;
    // beam lists of the WHOLE pixel set (the default pass shape: every pass covers every pixel), shared by the contexts: built once per tile set on the
    // stream of the pass that needs them first, the other contexts wait for evBeam once
    uint2 *beamEntries = nullptr, *beamCount = nullptr; size_t beamCap = 0; uint64_t beamGen = 0; hipEvent_t evBeam = nullptr;
    size_t ctxBytes() const { size_t b = beamCap * (KZ_BEAM_CAP + 1) * sizeof(uint2); for (const PassCtx &c : ctx) b += c.bytes(); return b; }
    KzPassInfo lastInfo{};
};
struct KzReplicaSet { std::mutex m; std::vector<KzDeviceState *> v; };

static KzReplicaSet *replicaSet(const KzScene *scene) { return (KzReplicaSet *)scene->dev; }

template <class Tp> static int uploadVec(KzDeviceState *ds, const std::vector<Tp> &v, const Tp **out) {
    *out = nullptr;
    void *p = nullptr;
    const size_t bytes = v.empty() ? 256 : v.size() * sizeof(Tp);    // keep a valid (dummy) pointer so kernels never see null
    KZ_ALLOC(&p, bytes);
    ds->allocs.push_back(p);
    if (v.empty()) HIP_TRY(hipMemset(p, 0, bytes));
    else HIP_TRY(hipMemcpy(p, v.data(), bytes, hipMemcpyHostToDevice));
    *out = (const Tp *)p;
    return KZ_OK;
}

static void releaseReplica(KzDeviceState *ds) {
    (void)hipSetDevice(ds->device);
    (void)hipDeviceSynchronize();
    for (void *p : ds->allocs) (void)hipFree(p);
    for (void *p : {(void *)ds->film, (void *)ds->srgb, (void *)ds->pixList, (void *)ds->pixIndex, (void *)ds->stats, (void *)ds->packDev, (void *)ds->rectsDev, (void *)ds->prevDev, (void *)ds->beamEntries, (void *)ds->beamCount}) if (p) (void)hipFree(p);
    if (ds->evBeam) (void)hipEventDestroy(ds->evBeam);
    if (ds->packHost) (void)hipHostFree(ds->packHost);
    for (auto &c : ds->ctx) { c.release(); for (auto &e : c.stageEv) (void)hipEventDestroy(e); }
    for (auto &e : ds->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipStream_t st : ds->passStream) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : ds->evFilm) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ds->evFork, ds->evCallA, ds->evCallB}) if (e) (void)hipEventDestroy(e);
    delete ds;
}

void kz_device_init(KzScene *scene) { scene->dev = new KzReplicaSet(); }

void kz_device_release(KzScene *scene) {
    KzReplicaSet *rs = replicaSet(scene);
    if (!rs) return;
    for (KzDeviceState *ds : rs->v) releaseReplica(ds);
    delete rs;
    scene->dev = nullptr;
}

// KzTuning -> the kernels' KzTune. Zero = the library default. (The KZ_* environment overrides of ABI v2 / v3 are gone: nothing in the
// library reads the environment.) Fields that select a kernel of kz_experiments.h are honoured only by a -DKZ_EXPERIMENTS build; the
// default library refuses them loudly instead of ignoring them.
static int resolveTune(const KzTuning &t, KzTune &r) {
    auto pick = [](int a, int d) { return a > 0 ? a : d; };
    r = KzTune{};
    r.refill = pick(t.refill, 0); r.postpone = pick(t.postpone, 24); r.batch = pick(t.batch, 0);                      // 0: the default per ray kind (wfPass)
    r.travBlocksPerCU = std::min(8, pick(t.traceBlocksPerCU, KZ_TRACE_WAVES)); r.shadeBlocksPerCU = std::min(16, pick(t.shadeBlocksPerCU, 0));
    r.ldsStack = pick(t.ldsStack, 16);
    r.packet = pick(t.packetPrimary, 0); r.filmGather = pick(t.filmGather, 0);
#if defined(KZ_EXPERIMENTS) && defined(KZ_SHADE_SPLIT)
    r.shadeSplit = KZ_SHADE_SPLIT;          // (development builds only: -DKZ_EXPERIMENTS -DKZ_SHADE_SPLIT=1, profiles/r04b_shade_split)
#endif
    r.wide = t.bvh2 ? 0 : 1; r.keyStack = pick(t.keyStack, 0); r.ldsTop = pick(t.ldsTop, 0); r.leafQueue = pick(t.leafQueue, 0);
    r.legacyTrace = pick(t.legacyTrace, 0); r.mixed = pick(t.mixedLaunch, 0);
#ifndef KZ_EXPERIMENTS
    if (t.bvh2 || t.keyStack > 0 || t.ldsTop > 0 || t.leafQueue > 1 || t.legacyTrace > 0 || t.mixedLaunch > 0)
        return kz_fail(KZ_ERR_UNSUPPORTED, "KzTuning.bvh2 / keyStack / ldsTop / leafQueue / legacyTrace / mixedLaunch select kernels of rejected experiments: "
                                           "this library was built without -DKZ_EXPERIMENTS (kz_build_flags)");
#endif
    r.ovf = nullptr; r.ovfStride = 0;
    return KZ_OK;
}

static int findReplica(const KzScene *scene, int device, KzDeviceState **out) {
    if (!scene) return kz_fail(KZ_ERR_INVALID_ARG, "null scene");
    KzReplicaSet *rs = replicaSet(scene);
    KzDeviceState *ds = nullptr;
    if (rs) {
        std::lock_guard<std::mutex> g(rs->m);
        if (device < 0) ds = rs->v.empty() ? nullptr : rs->v.front();
        else for (KzDeviceState *d : rs->v) if (d->device == device) { ds = d; break; }
    }
    if (!ds) {
        if (device < 0 || !rs || rs->v.empty()) return kz_fail(KZ_ERR_STATE, "scene is not on a device: call kz_scene_upload first");
        return kz_fail(KZ_ERR_STATE, "scene is not resident on device %d: call kz_scene_upload(scene, %d) first", device, device);
    }
    hipError_t e = hipSetDevice(ds->device);
    if (e != hipSuccess) return kz_fail(KZ_ERR_HIP, "hipSetDevice(%d): %s", ds->device, hipGetErrorString(e));
    *out = ds;
    return KZ_OK;
}
// the primary replica (calls without a device argument)
static int requireDevice(KzScene *scene, KzDeviceState **out) { return findReplica(scene, -1, out); }

static int uploadReplica(KzScene *scene, KzDeviceState *ds) {
    int rc;
    if ((rc = uploadVec(ds, scene->nodes, &ds->T.nodes))) return rc;
    if ((rc = uploadVec(ds, scene->nodes4, &ds->T.nodes4))) return rc;
    if ((rc = uploadVec(ds, scene->tris, &ds->T.tris))) return rc;
    if ((rc = uploadVec(ds, scene->shade, &ds->T.shade))) return rc;
    if ((rc = uploadVec(ds, scene->meshRows, &ds->T.meshes))) return rc;
    if ((rc = uploadVec(ds, scene->bsdfs, &ds->T.bsdfs))) return rc;
    if ((rc = uploadVec(ds, scene->lightRows, &ds->T.lights))) return rc;
    if ((rc = uploadVec(ds, scene->cdf, &ds->T.cdf))) return rc;
    if ((rc = uploadVec(ds, scene->pmj, &ds->T.pmj))) return rc;
    if ((rc = uploadVec(ds, scene->bn, &ds->T.bn))) return rc;
    if ((rc = uploadVec(ds, scene->pixelSamples, &ds->T.pixelSamples))) return rc;
    if ((rc = uploadVec(ds, scene->jump, &ds->T.jump))) return rc;
    std::vector<float> ft(scene->filter, scene->filter + KZ_FILTER_RESOLUTION + 1);
    if ((rc = uploadVec(ds, ft, &ds->T.filter))) return rc;
    if ((rc = uploadVec(ds, scene->ilTris, &ds->T.ilTris))) return rc;
    if ((rc = uploadVec(ds, scene->texProgs, &ds->T.texProgs))) return rc;
    if ((rc = uploadVec(ds, scene->texOps, &ds->T.texOps))) return rc;
    if ((rc = uploadVec(ds, scene->images, &ds->T.images))) return rc;
    if ((rc = uploadVec(ds, scene->texels, &ds->T.texels))) return rc;
    const KzParams &P = scene->prm;
    ds->filmPixels = (size_t)(P.width + 2 * P.border) * (size_t)(P.height + 2 * P.border);
    KZ_ALLOC(&ds->film, ds->filmPixels * sizeof(float4));
    HIP_TRY(hipMemset(ds->film, 0, ds->filmPixels * sizeof(float4)));
    KZ_ALLOC(&ds->stats, 32 * sizeof(unsigned long long));             // 8 counters of KzStats + 16 lane statistics of the -DKZ_LANESTAT development build + 3 beam-list counters
    HIP_TRY(hipMemset(ds->stats, 0, 32 * sizeof(unsigned long long)));
    { hipDeviceProp_t prop; HIP_TRY(hipGetDeviceProperties(&prop, ds->device)); ds->numCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256; ds->totalMem = prop.totalGlobalMem; }
    HIP_TRY(hipFuncSetAttribute((const void *)kz_film_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    HIP_TRY(hipDeviceSynchronize());
    return KZ_OK;
}

extern "C" {

void kz_debug_fail_alloc(int nth) { g_failAlloc = nth > 0 ? nth : 0; }

int kz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int kz_device_mem_info(int device, uint64_t *freeBytes, uint64_t *totalBytes) {
    int n = kz_device_count();
    if (device < 0 || device >= n) return kz_fail(n ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (freeBytes) *freeBytes = f;
    if (totalBytes) *totalBytes = t;
    return KZ_OK;
}

// random::permute on the device (the function the sampler kernels call), for the known-answer vectors minted from the reference's own text
int kz_debug_permute(int device, uint32_t n, const uint32_t *i, const uint32_t *l, const uint32_t *p, uint32_t *out) {
    int nd = kz_device_count();
    if (device < 0 || device >= nd) return kz_fail(nd ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, nd);
    if (!n) return KZ_OK;
    if (!i || !l || !p || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(device));
    DevMem dI, dL, dP, dO;
    const size_t bytes = (size_t)n * sizeof(uint32_t);
    KZ_ALLOC(&dI.p, bytes); KZ_ALLOC(&dL.p, bytes); KZ_ALLOC(&dP.p, bytes); KZ_ALLOC(&dO.p, bytes);
    HIP_TRY(hipMemcpy(dI.p, i, bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dL.p, l, bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dP.p, p, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kz_permute_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, dI.as<uint32_t>(), dL.as<uint32_t>(), dP.as<uint32_t>(), dO.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dO.p, bytes, hipMemcpyDeviceToHost));
    return KZ_OK;
}

int kz_debug_exact_math_check(int device, uint64_t *rcpMismatches, uint64_t *sqrtMismatches, uint64_t *checked) {
    int n = kz_device_count();
    if (device < 0 || device >= n) return kz_fail(n ? KZ_ERR_INVALID_ARG : KZ_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n);
    HIP_TRY(hipSetDevice(device));
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

int kz_scene_upload(KzScene *scene, int device) {
    if (!scene) return kz_fail(KZ_ERR_INVALID_ARG, "null scene");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return kz_fail(KZ_ERR_NO_DEVICE, "no HIP device visible (the product path has no CPU fallback)");
    if (device < 0 || device >= n) return kz_fail(KZ_ERR_INVALID_ARG, "device %d out of range (%d visible)", device, n);
    KzReplicaSet *rs = replicaSet(scene);
    {
        std::lock_guard<std::mutex> g(rs->m);
        for (KzDeviceState *d : rs->v) if (d->device == device) return KZ_OK;       // already resident
    }
    HIP_TRY(hipSetDevice(device));
    KzDeviceState *ds = new KzDeviceState();
    ds->device = device;
    const int rc = uploadReplica(scene, ds);
    if (rc) { releaseReplica(ds); return rc; }
    std::lock_guard<std::mutex> g(rs->m);
    for (KzDeviceState *d : rs->v) if (d->device == device) { releaseReplica(ds); return KZ_OK; }     // lost a race for the same device
    rs->v.push_back(ds);
    return KZ_OK;
}

int kz_scene_evict(KzScene *scene, int device) {
    if (!scene) return kz_fail(KZ_ERR_INVALID_ARG, "null scene");
    KzReplicaSet *rs = replicaSet(scene);
    std::vector<KzDeviceState *> gone;
    {
        std::lock_guard<std::mutex> g(rs->m);
        for (size_t i = 0; i < rs->v.size();) {
            if (device < 0 || rs->v[i]->device == device) { gone.push_back(rs->v[i]); rs->v.erase(rs->v.begin() + i); } else ++i;
        }
    }
    if (gone.empty() && device >= 0) return kz_fail(KZ_ERR_STATE, "scene is not resident on device %d", device);
    for (KzDeviceState *d : gone) releaseReplica(d);
    return KZ_OK;
}

int kz_scene_devices(const KzScene *scene, int32_t *devices, uint32_t cap, uint32_t *count) {
    if (!scene || !count) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    KzReplicaSet *rs = replicaSet(scene);
    std::lock_guard<std::mutex> g(rs->m);
    *count = (uint32_t)rs->v.size();
    for (uint32_t i = 0; i < *count && i < cap && devices; ++i) devices[i] = rs->v[i]->device;
    return KZ_OK;
}

} // extern "C"

// pixel list (8x8 blocks row-major inside each tile, row-major inside a block) + image-sized index map
static int prepareTiles(KzScene *scene, KzDeviceState *ds, const KzTile *tiles, uint32_t nTiles, hipStream_t stream) {
    const KzParams &P = scene->prm;
    KzTile whole = {0, 0, P.width, P.height};
    if (!tiles || nTiles == 0) { tiles = &whole; nTiles = 1; }
    bool same = ds->tilesValid && ds->curTiles.size() == nTiles && std::memcmp(ds->curTiles.data(), tiles, nTiles * sizeof(KzTile)) == 0;
    if (same) return KZ_OK;
    std::vector<uint32_t> list;
    std::vector<int32_t> index((size_t)P.width * P.height, -1);
    for (uint32_t t = 0; t < nTiles; ++t) {
        const KzTile &tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.w <= 0 || tl.h <= 0 || tl.x0 + tl.w > P.width || tl.y0 + tl.h > P.height)
            return kz_fail(KZ_ERR_INVALID_ARG, "tile %u (%d,%d %dx%d) outside the %dx%d image", t, tl.x0, tl.y0, tl.w, tl.h, P.width, P.height);
        for (int by = tl.y0; by < tl.y0 + tl.h; by += 8)
            for (int bx = tl.x0; bx < tl.x0 + tl.w; bx += 8)
                for (int y = by; y < std::min(by + 8, tl.y0 + tl.h); ++y)
                    for (int x = bx; x < std::min(bx + 8, tl.x0 + tl.w); ++x) {
                        int32_t &slot = index[(size_t)y * P.width + x];
                        if (slot >= 0) return kz_fail(KZ_ERR_INVALID_ARG, "tiles overlap at pixel (%d,%d)", x, y);
                        slot = (int32_t)list.size();
                        list.push_back((uint32_t)x | ((uint32_t)y << 16));
                    }
    }
    HIP_TRY(hipStreamSynchronize(stream));
    for (hipStream_t st : ds->passStream) if (st) HIP_TRY(hipStreamSynchronize(st));
    ds->tilesValid = false;
    if (!ds->pixIndex) KZ_ALLOC(&ds->pixIndex, index.size() * sizeof(int32_t));
    if (list.size() > ds->pixCap) {
        if (ds->pixList) (void)hipFree(ds->pixList);
        ds->pixList = nullptr; ds->pixCap = 0;
        KZ_ALLOC(&ds->pixList, list.size() * sizeof(uint32_t));
        ds->pixCap = list.size();
    }
    HIP_TRY(hipMemcpy(ds->pixList, list.data(), list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ds->pixIndex, index.data(), index.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    ds->nPix = (uint32_t)list.size();
    ds->curTiles.assign(tiles, tiles + nTiles);
    ds->tilesValid = true; ++ds->tileGen;
    return KZ_OK;
}

static int stageMark(PassCtx &c, hipStream_t stream, int kind) {
    if (c.stageUsed == c.stageEv.size()) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); c.stageEv.push_back(e); c.stageKind.push_back(0); }
    c.stageKind[c.stageUsed] = kind;
    HIP_TRY(hipEventRecord(c.stageEv[c.stageUsed++], stream));
    return KZ_OK;
}

// Path state per (pixel, sample) item of a pass in flight: 8 float4 + uint4 + 3 queue words (wavefront) + 5 sample floats.
static constexpr size_t KZ_STATE_BYTES_PER_ITEM = 8 * sizeof(float4) + sizeof(uint4) + 3 * sizeof(uint32_t);
static constexpr size_t KZ_SAMPLE_BYTES_PER_ITEM = 5 * sizeof(float);
static constexpr size_t KZ_TAP_BYTES_PER_PIXEL = (size_t)KZ_TAPS_MAX * KZ_TAPS_MAX * sizeof(float4);

// ---- buffers of one pass context: sized for `need` items of `nPix` pixels; nothing is left half-allocated on failure ----
static int ctxEnsure(PassCtx &c, size_t need, size_t nPix, bool wavefront, bool tapSums, bool beams, hipStream_t stream) {
    if (beams && nPix > c.beamCap) {                                // (beams: lists of this context's own pixel chunks; the whole-set lists live with the replica)
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.beamEntries) (void)hipFree(c.beamEntries);
        if (c.beamCount) (void)hipFree(c.beamCount);
        c.beamEntries = nullptr; c.beamCount = nullptr; c.beamCap = 0; c.beamGen = 0;
        KZ_ALLOC(&c.beamEntries, nPix * KZ_BEAM_CAP * sizeof(uint2));
        KZ_ALLOC(&c.beamCount, nPix * sizeof(uint2));
        c.beamCap = nPix;
    }
    if (tapSums && nPix > c.tapsCap) {                               // (only the tap-sum film path has this buffer)
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.taps) (void)hipFree(c.taps);
        c.taps = nullptr; c.tapsCap = 0;
        KZ_ALLOC(&c.taps, nPix * KZ_TAP_BYTES_PER_PIXEL);
        c.tapsCap = nPix;
    }
    if (need > c.sampCap) {
        HIP_TRY(hipStreamSynchronize(stream));
        if (c.samp) (void)hipFree(c.samp);
        c.samp = nullptr; c.sampCap = 0;
        KZ_ALLOC(&c.samp, need * KZ_SAMPLE_BYTES_PER_ITEM);
        c.sampCap = need;
    }
    if (wavefront && (need > c.wfCap || c.wfAllocs.empty())) {
        HIP_TRY(hipStreamSynchronize(stream));
        for (void *p : c.wfAllocs) (void)hipFree(p);
        c.wfAllocs.clear(); c.wfCap = 0; c.wf = KzWf{};
        KzWf W{};
        auto alloc = [&](void **p, size_t bytes) -> int { KZ_ALLOC(p, bytes); c.wfAllocs.push_back(*p); return KZ_OK; };
        int rc = KZ_OK;
        if (KZ_STATE_AOS) {                                             // two 64-B records per slot: (rayA rayB hit thr) and (shA shB shL misc)
            float4 *rec[2] = {nullptr, nullptr};
            for (int k = 0; k < 2; ++k) if (!rc) rc = alloc((void **)&rec[k], need * 4 * sizeof(float4));
            if (!rc) { W.rayA.p = rec[0]; W.rayB.p = rec[0] + 1; W.hit.p = rec[0] + 2; W.thr.p = rec[0] + 3; W.shA.p = rec[1]; W.shB.p = rec[1] + 1; W.shL.p = rec[1] + 2; W.misc.p = rec[1] + 3; }
        } else
            for (KzField<float4> *f : {&W.rayA, &W.rayB, &W.hit, &W.thr, &W.misc, &W.shA, &W.shB, &W.shL}) if (!rc) rc = alloc((void **)&f->p, need * sizeof(float4));
        if (!rc) rc = alloc((void **)&W.smp, need * sizeof(uint4));
        for (int q = 0; q < 3; ++q) if (!rc) rc = alloc((void **)&W.queue[q], need * sizeof(uint32_t));
        if (!rc) rc = alloc((void **)&W.counts, 8 * 520 * sizeof(uint32_t));
        if (rc) { for (void *p : c.wfAllocs) (void)hipFree(p); c.wfAllocs.clear(); return rc; }
        c.wf = W; c.wfCap = need;
    }
    return KZ_OK;
}

#ifdef KZ_EXPERIMENTS
// Launch code of the kernels of kz_experiments.h (development builds only): takes over a traversal launch of wfPass when the caller's
// KzTuning selects one of the rejected experiments. The films stay bit-identical to the product kernels' (tests/test_gpu_configs.py).
struct KzExpLaunch {
    KzScene *scene; KzDeviceState *ds; PassCtx *c; hipStream_t stream; KzWf W; KzTune tune; dim3 gTrav; size_t traceLds; int stackBound; bool st; uint32_t items;
    bool keys = false, dq = false, any = false; size_t ldsX = 0, dqLds = 0, stackBytes = 0; KzTune tuneDq{};
    int prepare() {
        const KzParams &P = scene->prm;
        keys = tune.wide && tune.keyStack == 2;
        dq = tune.wide && tune.leafQueue == 2;
        tune.ldsTop = tune.wide ? (int)std::min<size_t>((size_t)std::max(0, tune.ldsTop), std::min<size_t>(scene->nodes4.size(), 1536)) : 0;
        any = !tune.wide || keys || tune.ldsTop > 0 || dq || tune.legacyTrace || tune.mixed;
        if (!any) return KZ_OK;
        stackBytes = (size_t)P.stackDepth * KZ_BLOCK * sizeof(uint32_t);
        ldsX = (size_t)(tune.ldsStack + 1) * KZ_BLOCK * sizeof(uint32_t) * (keys ? 2 : 1) + (size_t)tune.ldsTop * sizeof(KzNode4);
        if (ldsX > 64 * 1024) {
            HIP_TRY(hipFuncSetAttribute((const void *)kz_wf_trace_x<0, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_TRY(hipFuncSetAttribute((const void *)kz_wf_trace_x<2, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        if (dq) {
            const int dqLS = std::max(2, std::min(tune.ldsStack, std::min(stackBound, 9)));             // 9 rows + queue + results = 19.4 KB per workgroup: 8 per CU
            dqLds = (size_t)4 * ((size_t)(dqLS + 1) * 64 + 128 + 192 + 2 * KZ_DQ_JOBS) * sizeof(uint32_t);
            tuneDq = tune; tuneDq.ldsStack = dqLS;
            const size_t stride = (size_t)gTrav.x * KZ_BLOCK, needOvf = stride * (size_t)std::max(1, stackBound - dqLS);
            if (needOvf > c->ovfCap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (c->ovf) (void)hipFree(c->ovf);
                c->ovf = nullptr; c->ovfCap = 0;
                KZ_ALLOC(&c->ovf, needOvf * sizeof(uint32_t));
                c->ovfCap = needOvf;
                tune.ovf = c->ovf;
            }
            tuneDq.ovf = c->ovf; tuneDq.ovfStride = (uint32_t)stride;
            if (items > c->litCap) {
                HIP_TRY(hipStreamSynchronize(stream));
                if (c->litQueue) (void)hipFree(c->litQueue);
                c->litQueue = nullptr; c->litCap = 0;
                KZ_ALLOC(&c->litQueue, (size_t)items * sizeof(uint32_t));
                c->litCap = items;
            }
        }
        return KZ_OK;
    }
    bool allowsPacket() const { return tune.wide && !tune.legacyTrace; }
    // the round-2 kernel with its options; MODE 0, 1, 2, 3
    template <int MODE> void traceX(const uint32_t *q, const uint32_t *cptr, uint32_t cimm, uint32_t *head, const uint32_t *qb, const uint32_t *cb) {
        const KzParams &P = scene->prm; const dim3 blk(KZ_BLOCK);
        constexpr int M = MODE;
        if (tune.wide && tune.ldsTop > 0 && !st && (M == 0 || M == 2)) hipLaunchKernelGGL((kz_wf_trace_x<(M == 0 || M == 2) ? M : 0, false, true, false, true>), gTrav, blk, ldsX, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        else if (tune.wide && keys && (M == 0 || M == 1)) {
            if (st) hipLaunchKernelGGL((kz_wf_trace_x<(M == 0 || M == 1) ? M : 0, true, true, true>), gTrav, blk, ldsX, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
            else hipLaunchKernelGGL((kz_wf_trace_x<(M == 0 || M == 1) ? M : 0, false, true, true>), gTrav, blk, ldsX, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        } else if (tune.wide) {
            if (st) hipLaunchKernelGGL((kz_wf_trace_x<M, true, true>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
            else hipLaunchKernelGGL((kz_wf_trace_x<M, false, true>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        } else {
            if (st) hipLaunchKernelGGL((kz_wf_trace_x<M, true, false>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
            else hipLaunchKernelGGL((kz_wf_trace_x<M, false, false>), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, tune, qb, cb);
        }
    }
    // closest-hit launches outside the bounce loop (camera rays, first-hit walk-through) and the lit-ray walk-through
    bool trace(int mode, const uint32_t *q, const uint32_t *cptr, uint32_t cimm, uint32_t *head, uint32_t *qb, uint32_t *cb) {
        if (!any) return false;
        const KzParams &P = scene->prm; const dim3 blk(KZ_BLOCK);
        if (tune.legacyTrace && (mode == 0 || mode == 1)) {
            if (mode == 0) { if (st) hipLaunchKernelGGL((kz_wf_extend<true, false>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); else hipLaunchKernelGGL((kz_wf_extend<false, false>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); }
            else { if (st) hipLaunchKernelGGL((kz_wf_extend<true, true>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); else hipLaunchKernelGGL((kz_wf_extend<false, true>), gTrav, blk, stackBytes, stream, P, ds->T, W, q, cptr, cimm); }
            return true;
        }
        if (mode == 0) { traceX<0>(q, cptr, cimm, head, qb, cb); return true; }
        if (mode == 1) { traceX<1>(q, cptr, cimm, head, qb, cb); return true; }
        if (mode == 2) { traceX<2>(q, cptr, cimm, head, qb, cb); return true; }
        return false;
    }
    // the traversal launches of one bounce (shadow rays of this bounce, closest-hit rays of the next)
    bool bounce(int iter, bool needExtend, uint32_t *nextQ, uint32_t *nextCount, uint32_t *shQ, uint32_t *shCount) {
        if (!any) return false;
        const KzParams &P = scene->prm; const dim3 blk(KZ_BLOCK);
        if (!tune.legacyTrace && tune.mixed && P.nLights > 0 && needExtend) {
            traceX<3>(nextQ, nextCount, 0u, nextCount + 2, shQ, shCount);
            (void)stageMark(*c, stream, 1);
            return true;
        }
        if (P.nLights > 0) {
            if (dq && P.shadowFast) {
                uint32_t *litCount = W.counts + 4 * 520 + 2 * (iter + 1), *litHead = litCount + 1;
                if (st) hipLaunchKernelGGL((kz_wf_trace_dq<2, true>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount, 0u, nextCount + 3, tuneDq, c->litQueue, litCount);
                else hipLaunchKernelGGL((kz_wf_trace_dq<2, false>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount, 0u, nextCount + 3, tuneDq, c->litQueue, litCount);
                traceX<2>(c->litQueue, litCount, 0u, litHead, nullptr, nullptr);      // the few rays that cross an invisible light
            } else if (tune.legacyTrace) {
                if (st) hipLaunchKernelGGL(kz_wf_shadow<true>, gTrav, blk, stackBytes, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount);
                else hipLaunchKernelGGL(kz_wf_shadow<false>, gTrav, blk, stackBytes, stream, P, ds->T, W, (const uint32_t *)shQ, (const uint32_t *)shCount);
            } else traceX<2>(shQ, shCount, 0u, nextCount + 3, nullptr, nullptr);
        }
        (void)stageMark(*c, stream, 3);
        if (needExtend) {
            if (dq && st) hipLaunchKernelGGL((kz_wf_trace_dq<0, true>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)nextQ, (const uint32_t *)nextCount, 0u, nextCount + 2, tuneDq, (uint32_t *)nullptr, (uint32_t *)nullptr);
            else if (dq) hipLaunchKernelGGL((kz_wf_trace_dq<0, false>), gTrav, blk, dqLds, stream, P, ds->T, W, (const uint32_t *)nextQ, (const uint32_t *)nextCount, 0u, nextCount + 2, tuneDq, (uint32_t *)nullptr, (uint32_t *)nullptr);
            else trace(0, nextQ, nextCount, 0u, nextCount + 2, nullptr, nullptr);
            (void)stageMark(*c, stream, 1);
        }
        return true;
    }
};
#endif

// kz_wf_trace instantiations by (mode, stats): the launch code picks from this table instead of a ladder of macros
typedef void (*KzTraceFn)(KzParams, KzDevTables, KzWf, const uint32_t *, const uint32_t *, uint32_t, uint32_t *, KzTune, uint32_t *, uint32_t *);
static KzTraceFn traceFn(int mode, bool stats) {
    static const KzTraceFn tab[5][2] = {{kz_wf_trace<0, false>, kz_wf_trace<0, true>}, {kz_wf_trace<1, false>, kz_wf_trace<1, true>},
                                        {kz_wf_trace<2, false>, kz_wf_trace<2, true>}, {nullptr, nullptr}, {kz_wf_trace<4, false>, kz_wf_trace<4, true>}};
    return tab[mode][stats ? 1 : 0];
}

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

// One pass of the wavefront pipeline over `items` = nPixPass x Sp (pixel, sample) items: pixels pixList[0 .. nPixPass), sample indices
// [sBegin, sBegin + Sp). Every launch goes to `stream`; queue counts stay on the device.
static int wfPass(KzScene *scene, KzDeviceState *ds, PassCtx &c, hipStream_t stream, const uint32_t *pixList, uint32_t p0, uint32_t nPixPass, uint32_t sBegin, uint32_t Sp, uint32_t items, KzTune tune, bool beams) {
    const KzParams &P = scene->prm;
    KzWf W = c.wf;
    W.outJx = c.samp; W.outJy = c.samp + c.sampCap; W.outR = c.samp + 2 * c.sampCap; W.outG = c.samp + 3 * c.sampCap; W.outB = c.samp + 4 * c.sampCap; W.stats = ds->stats;
    const bool st = ds->statsOn;
    const dim3 blk(KZ_BLOCK);
    // shade: the lean variant runs 4 workgroups per CU at once (its launch bounds), so the default grid is exactly those, each looping over
    // its share: with 6 per CU the kernel ran 1.5 rounds, the second with half the CUs idle (same-call sweep, profiles/r02h_shade: shade alone
    // 24.1 ms at 6, 23.0 at 4, 22.6 at 8, 22.3 at 12 - and with two passes in flight the small grid leaves the most room for the other pass:
    // 1246 Msamples/s at 4 against 1214-1218 at 6 / 8 / 12). The EXT variant (3 resident) shows no preference on C3 and stays at 6.
    const int shadeBlocks = tune.shadeBlocksPerCU > 0 ? tune.shadeBlocksPerCU : (P.bsdfExt ? 6 : KZ_SHADE_WAVES);
    const dim3 gTrav((unsigned)(ds->numCU * tune.travBlocksPerCU)), gShade((unsigned)(ds->numCU * shadeBlocks));
    const dim3 gPacket((unsigned)(ds->numCU * 8));                  // the packet kernel is compiled for 8 waves per SIMD whatever KZ_TRACE_WAVES is
    // stack: tune.ldsStack entries per lane in LDS, the rest of the worst case (known from the builder) in a global overflow area
    const int stackBound = std::max(tune.wide ? P.stackBound4 : P.stackDepth, 2);
    tune.ldsStack = std::max(2, std::min(tune.ldsStack, stackBound));
    const size_t traceLds = (size_t)(tune.ldsStack + 1) * KZ_BLOCK * sizeof(uint32_t);      // + one scratch slot per lane (branch-free pushes)
    {
        const size_t stride = (size_t)gTrav.x * KZ_BLOCK, needOvf = stride * (size_t)std::max(1, stackBound - tune.ldsStack) * 2;      // (x 2: the key stack of kz_experiments.h)
        if (needOvf > c.ovfCap) {
            HIP_TRY(hipStreamSynchronize(stream));
            if (c.ovf) (void)hipFree(c.ovf);
            c.ovf = nullptr; c.ovfCap = 0;
            KZ_ALLOC(&c.ovf, needOvf * sizeof(uint32_t));
            c.ovfCap = needOvf;
        }
        tune.ovf = c.ovf; tune.ovfStride = (uint32_t)stride;
    }
    const int maxDepth = P.maxDepth;
    c.stageUsed = 0;
    HIP_TRY(hipMemsetAsync(W.counts, 0, 8 * 520 * sizeof(uint32_t), stream));
    { int rc_ = stageMark(c, stream, -1); if (rc_) return rc_; }
    hipLaunchKernelGGL(kz_wf_generate, dim3((items + KZ_BLOCK - 1) / KZ_BLOCK), blk, 0, stream, P, ds->T, W, pixList, items, Sp, sBegin);
    { int rc_ = stageMark(c, stream, 0); if (rc_) return rc_; }
    if (maxDepth <= 0) return KZ_OK;           // Li returns 0 before the loop contributes anything
#ifdef KZ_EXPERIMENTS
    KzTune tuneX = tune; if (tuneX.batch <= 0) tuneX.batch = 128; if (tuneX.refill <= 0) tuneX.refill = 40;
    KzExpLaunch X{scene, ds, &c, stream, W, tuneX, gTrav, traceLds, stackBound, st, items};
    if (int rc_ = X.prepare()) return rc_;
#endif
    // mode 0 / 1 / 2 / 4 of kz_wf_trace on queue q (nullptr: identity) of *cptr (nullptr: cimm) entries
    auto trace = [&](int mode, const uint32_t *q, const uint32_t *cptr, uint32_t cimm, uint32_t *head, uint32_t *qb, uint32_t *cb) {
#ifdef KZ_EXPERIMENTS
        if (X.trace(mode, q, cptr, cimm, head, qb, cb)) return;
#endif
        KzTune t = tune;
        if (t.batch <= 0) t.batch = 128;                 // (the least a wave reserves per global atomic: kz_wf_trace asks for more while much is left)
        // idle lanes are refilled once fewer than this many are busy. A refill of the shadow kernel is the dearer one (the invisible-light test of every new
        // ray), so it waits for more idle lanes: same-call sweep, shadow stage 21.7 / 20.9 / 20.9 ms at 40 / 32 / 28 on C3, 20.3 / 20.0 / 20.3 on C4; the
        // closest-hit kernel 32.85 / 33.3 / 34.2 on C4
        if (t.refill <= 0) t.refill = (mode == 2 || mode == 4) ? 32 : 40;
        hipLaunchKernelGGL(traceFn(mode, st), gTrav, blk, traceLds, stream, P, ds->T, W, q, cptr, cimm, head, t, qb, cb);
    };
    // camera rays: pixel beams + per-sample triangle tests (kz_wf_beam / kz_wf_trace_list), the wave-level packet traversal
    // (kz_wf_trace_packet) for what the beams cannot take, or the per-lane kernel on request
    bool packet = tune.packet != 1 && P.stackBound4 <= 128;
#ifdef KZ_EXPERIMENTS
    if (!X.allowsPacket()) packet = false;
#endif
#define KZ_PACKET(ST, FX, q, cptr, cimm, headp) hipLaunchKernelGGL((kz_wf_trace_packet<ST, FX>), gPacket, blk, 0, stream, P, ds->T, W, q, cptr, cimm, headp, 8, W.queue[2], W.counts + 0)
#define KZ_PACKET4(q, cptr, cimm, headp) do { if (P.anyInvisibleLight) { if (st) KZ_PACKET(true, true, q, cptr, cimm, headp); else KZ_PACKET(false, true, q, cptr, cimm, headp); } \
                                              else { if (st) KZ_PACKET(true, false, q, cptr, cimm, headp); else KZ_PACKET(false, false, q, cptr, cimm, headp); } } while (0)
    if (packet && beams) {
        const uint2 *lstEntries, *lstHeads;
        const int LS = std::max(1, std::min(P.stackBound4, KZ_BEAM_STACK));      // a beam whose open set would grow beyond this leaves the rest unexplored (t_valid)
        const dim3 gBeam((nPixPass + KZ_BLOCK - 1) / KZ_BLOCK);
        const size_t beamLds = (size_t)2 * LS * KZ_BLOCK * sizeof(uint32_t);
        if (p0 == 0 && nPixPass == ds->nPix) {                                 // the whole pixel set: ONE list for all contexts of the replica
            if (ds->beamGen != ds->tileGen) {                                   // (a call's pass streams all start behind the previous call: nobody reads the old lists any more)
                hipLaunchKernelGGL(kz_wf_beam, gBeam, blk, beamLds, stream, P, ds->T, pixList, nPixPass, LS, ds->beamEntries, ds->beamCount);
                if (st) hipLaunchKernelGGL(kz_wf_beam_count, gBeam, blk, 0, stream, (const uint2 *)ds->beamCount, nPixPass, ds->stats + 24);
                HIP_TRY(hipEventRecord(ds->evBeam, stream));
                ds->beamGen = ds->tileGen; c.sharedSeen = ds->tileGen;
            } else if (c.sharedSeen != ds->tileGen) { HIP_TRY(hipStreamWaitEvent(stream, ds->evBeam, 0)); c.sharedSeen = ds->tileGen; }
            lstEntries = ds->beamEntries; lstHeads = ds->beamCount;
        } else {                                                                // a pixel chunk: the context's own lists, built once per chunk
            if (c.beamGen != ds->tileGen || c.beamP0 != p0 || c.beamN != nPixPass) {
                hipLaunchKernelGGL(kz_wf_beam, gBeam, blk, beamLds, stream, P, ds->T, pixList, nPixPass, LS, c.beamEntries, c.beamCount);
                c.beamGen = ds->tileGen; c.beamP0 = p0; c.beamN = nPixPass;
                if (st) hipLaunchKernelGGL(kz_wf_beam_count, gBeam, blk, 0, stream, (const uint2 *)c.beamCount, nPixPass, ds->stats + 24);
            }
            lstEntries = c.beamEntries; lstHeads = c.beamCount;
        }
        uint32_t *fbQ = W.queue[0], *fbCount = W.counts + 8 * 520 - 8, *fbHead = fbCount + 1;      // rays of pixels whose list overflowed: the packet kernel's
        const dim3 gList((items + KZ_BLOCK - 1) / KZ_BLOCK);
#define KZ_LIST(ST, FX) hipLaunchKernelGGL((kz_wf_trace_list<ST, FX>), gList, blk, 0, stream, P, ds->T, W, items, Sp, lstEntries, lstHeads, fbQ, fbCount, W.queue[2], W.counts + 0)
        if (P.anyInvisibleLight) { if (st) KZ_LIST(true, true); else KZ_LIST(false, true); }
        else { if (st) KZ_LIST(true, false); else KZ_LIST(false, false); }
#undef KZ_LIST
        KZ_PACKET4((const uint32_t *)fbQ, (const uint32_t *)fbCount, 0u, fbHead);
    } else if (packet) {
        // (scenes with an invisible light: the epilogue queues the first hits on such a light for the walk-through launch below)
        KZ_PACKET4((const uint32_t *)nullptr, (const uint32_t *)nullptr, items, W.counts + 2);
    } else {
        trace(0, nullptr, nullptr, items, W.counts + 2, nullptr, nullptr);
        if (P.anyInvisibleLight) hipLaunchKernelGGL(kz_wf_primary_fix, gShade, blk, 0, stream, P, ds->T, W, items, W.queue[2], W.counts + 0);
    }
#undef KZ_PACKET4
#undef KZ_PACKET
    if (P.anyInvisibleLight) trace(1, W.queue[2], W.counts + 0, 0u, W.counts + 3, nullptr, nullptr);      // H6 walk-through of the first hit
    { int rc_ = stageMark(c, stream, 5); if (rc_) return rc_; }
    const uint32_t *cur = nullptr, *curCount = nullptr;
    const bool split = tune.shadeSplit != 0;
    const dim3 gClassify((unsigned)(ds->numCU * 8));
    for (int iter = 0; iter < maxDepth; ++iter) {
        // One kernel per bounce: the path queues ping-pong (W.queue[iter & 1] is written, the other one read). Two kernels (tune.shadeSplit): kz_wf_classify
        // reads the path queue and writes the survivors to W.queue[0]; kz_wf_shade_b reads those and writes the next path queue to W.queue[1], which
        // the classification has finished reading by then (one stream).
        uint32_t *nextQ = split ? W.queue[1] : W.queue[iter & 1], *nextCount = W.counts + 4 * (iter + 1), *shQ = W.queue[2], *shCount = W.counts + 4 * (iter + 1) + 1;
#ifdef KZ_EXPERIMENTS
        if (split) {
            uint32_t *svQ = W.queue[0], *svCount = W.counts + 6 * 520 + (iter + 1);
#define KZ_SHADE2(ST, EX) do { hipLaunchKernelGGL((kz_wf_classify<ST, EX>), gClassify, blk, 0, stream, P, ds->T, W, pixList, Sp, sBegin, iter, cur, curCount, items, svQ, svCount); \
                               hipLaunchKernelGGL((kz_wf_shade_b<ST, EX>), gShade, blk, 0, stream, P, ds->T, W, pixList, Sp, sBegin, iter, (const uint32_t *)svQ, (const uint32_t *)svCount, nextQ, nextCount, shQ, shCount); } while (0)
            if (st) { if (P.bsdfExt) KZ_SHADE2(true, true); else KZ_SHADE2(true, false); }
            else { if (P.bsdfExt) KZ_SHADE2(false, true); else KZ_SHADE2(false, false); }
#undef KZ_SHADE2
        } else
#endif
        {
#define KZ_SHADE(ST, EX) hipLaunchKernelGGL((kz_wf_shade<ST, EX>), gShade, blk, 0, stream, P, ds->T, W, pixList, Sp, sBegin, iter, cur, curCount, items, nextQ, nextCount, shQ, shCount)
            if (st) { if (P.bsdfExt) KZ_SHADE(true, true); else KZ_SHADE(true, false); }
            else { if (P.bsdfExt) KZ_SHADE(false, true); else KZ_SHADE(false, false); }
#undef KZ_SHADE
        }
        const bool lastIter = iter == maxDepth - 1;
        const bool needExtend = !lastIter || P.bgPresent;
#ifdef KZ_SORT_EXPERIMENT
        const uint32_t *trQ = nextQ, *trShQ = shQ;            // what the traversal launches read
        if (g_sortExp.on) {
            if (int rc_ = g_sortExp.ensure(scene, c.wfCap)) return rc_;
            uint32_t nB = 0, nS = 0;
            if (needExtend) { if (int rc_ = g_sortExp.sort(stream, W, 0, nextQ, nextCount, 0, &nB)) return rc_; trQ = g_sortExp.qOut[0]; }
            if (P.nLights > 0) { if (int rc_ = g_sortExp.sort(stream, W, 1, shQ, shCount, 1, &nS)) return rc_; trShQ = g_sortExp.qOut[1]; }
            if (iter == maxDepth - 1) { std::fprintf(stderr, "sortexp: bits %d dir %d: sorts of this pass %.3f ms\n", g_sortExp.bits, g_sortExp.useDir, g_sortExp.sortMs); g_sortExp.sortMs = 0; }
        }
#else
        const uint32_t *trQ = nextQ, *trShQ = shQ;
#endif
        { int rc_ = stageMark(c, stream, 2); if (rc_) return rc_; }
#ifdef KZ_EXPERIMENTS
        if (X.bounce(iter, needExtend, nextQ, nextCount, shQ, shCount)) { cur = nextQ; curCount = nextCount; continue; }
#endif
        if (P.nLights > 0) {
            if (P.shadowFast) {
                // any-hit kernel without the walk-through machinery; the (rare) rays whose segment crosses an invisible-light triangle go to a
                // queue - the ping-pong path queue this bounce's shade has just consumed - and are walked through by the general kernel
                uint32_t *litQ = split ? W.queue[0] : W.queue[(iter & 1) ^ 1], *litCount = W.counts + 4 * 520 + 2 * (iter + 1), *litHead = litCount + 1;
                trace(4, trShQ, shCount, 0u, nextCount + 3, litQ, litCount);
                if (P.anyInvisibleLight) trace(2, litQ, litCount, 0u, litHead, nullptr, nullptr);
            } else trace(2, trShQ, shCount, 0u, nextCount + 3, nullptr, nullptr);
        }
        { int rc_ = stageMark(c, stream, 3); if (rc_) return rc_; }
        if (needExtend) {
            trace(0, trQ, nextCount, 0u, nextCount + 2, nullptr, nullptr);
            int rc_ = stageMark(c, stream, 1); if (rc_) return rc_;
        }
        cur = nextQ; curCount = nextCount;
    }
    if (P.bgPresent) hipLaunchKernelGGL(kz_wf_final, gShade, blk, 0, stream, P, ds->T, W, cur, curCount);
    if (st) hipLaunchKernelGGL(kz_wf_count, dim3((items + KZ_BLOCK - 1) / KZ_BLOCK), blk, 0, stream, W, items);
    HIP_TRY(hipGetLastError());
    return KZ_OK;
}

static int renderOn(KzScene *scene, KzDeviceState *ds, const KzRenderOpts *opts) {
    int rc;
    const KzParams &P = scene->prm;
    if (opts->pipeline < 0 || opts->pipeline > 2) return kz_fail(KZ_ERR_INVALID_ARG, "pipeline %d (0 = default, 1 = megakernel, 2 = wavefront)", opts->pipeline);
    if (opts->passesInFlight < 0 || opts->passesInFlight > KZ_MAX_PASSES_IN_FLIGHT)
        return kz_fail(KZ_ERR_INVALID_ARG, "passesInFlight %d (0 = default, 1 .. %d)", opts->passesInFlight, KZ_MAX_PASSES_IN_FLIGHT);
    const int pipeline = opts->pipeline ? opts->pipeline : 2;
    uint32_t s0 = opts->sampleBegin, s1 = opts->sampleEnd;
    if (s0 == 0 && s1 == 0) s1 = P.sampleCount;
    if (s0 >= s1 || s1 > P.sampleCount) return kz_fail(KZ_ERR_INVALID_ARG, "sample range [%u,%u) outside [0,%u)", s0, s1, P.sampleCount);
    hipStream_t stream = (hipStream_t)opts->stream;
    ds->lastStream = stream;
    if ((rc = prepareTiles(scene, ds, opts->tiles, opts->nTiles, stream))) return rc;
    if (!opts->accumulate) HIP_TRY(hipMemsetAsync(ds->film, 0, ds->filmPixels * sizeof(float4), stream));
    KzTune tune;
    if ((rc = resolveTune(opts->tune, tune))) return rc;
    const int ftaps = P.tapHi - P.tapLo + 1;
    const bool tapSums = ftaps <= KZ_TAPS_MAX && tune.filmGather != 1;
    // ---- pass geometry. A pass is pixPerPass pixels x S samples of each = up to passItems (pixel, sample) items (default 2^27: 23.6 GB of
    // path state + sample records per pass in flight; 2^25 -> 981, 2^26 -> 1031, 2^27 -> 1061-1066, 2^28 -> 1071 Msamples/s on C4 in round 1:
    // fewer launches and shorter relative tails per sample). opts->tune.sppPerPass = 0: every pixel of the tile set and as many samples as
    // fit; n > 0: n samples (or all the call asks for) of as many pixels as fit, pixel chunks in the order of the pixel list.
    // The state never takes more than the caller's limit; without one, not more than 3/4 of the device and not more than what
    // is free now plus what this replica already holds for the purpose (another process or replica may own the rest).
    int nCtx = pipeline == 2 ? (opts->passesInFlight ? opts->passesInFlight : KZ_DEFAULT_PASSES_IN_FLIGHT) : 1;
    const size_t perItem = (pipeline == 2 ? KZ_STATE_BYTES_PER_ITEM : 0) + KZ_SAMPLE_BYTES_PER_ITEM;
    // camera rays by pixel beams: a pinhole camera with an affine sample map, a stack that fits LDS twice, unless the caller asks otherwise
    const bool beams = pipeline == 2 && P.beamOk && tune.packet != 1 && tune.packet != 2 && P.maxDepth > 0;
    const size_t perPixel = (tapSums ? KZ_TAP_BYTES_PER_PIXEL : 0) + (beams ? (KZ_BEAM_CAP + 1) * sizeof(uint2) : 0);
    size_t limit = opts->maxStateBytes;
    if (!limit) {
        size_t freeB = 0, totalB = 0;
        const size_t held = ds->ctxBytes();
        if (hipMemGetInfo(&freeB, &totalB) == hipSuccess && totalB > 0) limit = std::min(totalB / 4 * 3, freeB + held - std::min(freeB + held, (size_t)256 << 20));
        else limit = (size_t)32 << 30;
    }
    const uint32_t nSamples = s1 - s0;
    const size_t wantItems = std::max<size_t>(opts->passItems ? (size_t)opts->passItems : (size_t)1 << 27, 64);
    uint32_t S, pixPerPass;
    if (opts->tune.sppPerPass > 0) { S = std::min<uint32_t>((uint32_t)opts->tune.sppPerPass, nSamples); pixPerPass = (uint32_t)std::min<size_t>(ds->nPix, std::max<size_t>(64, wantItems / S / 64 * 64)); }
    else {
        pixPerPass = ds->nPix; S = (uint32_t)std::min<size_t>(std::max<size_t>(1, wantItems / std::max<uint32_t>(1, ds->nPix)), nSamples);
        // A frame too large for 64 samples of every pixel per pass (C5 on one GPU: 16) is rendered in pixel chunks of 256 samples instead: a wave of the
        // camera-ray kernels is then one pixel again (one shared list), the film stage touches a chunk per pass instead of the whole frame, and the paths of a pass
        // stay in a part of the scene (C5, same call: 1 586 Msamples/s at 16 x all pixels, 1 708 at 64 x 2 M, 1 734 at 256 x 512 K).
        if (S < 64 && nSamples >= 64) { S = std::min<uint32_t>(256u, nSamples); pixPerPass = (uint32_t)std::min<size_t>(ds->nPix, std::max<size_t>(64, wantItems / S / 64 * 64)); }
        // a multiple of 64 samples per pixel keeps every wave of the camera-ray kernels inside one pixel (one shared leaf list) - taken when it costs no extra pass
        // (a rank's share of a frame: 2^27 / 1 036 800 pixels = 129 -> 128)
        else if (S > 64 && S % 64 && (nSamples + S / 64 * 64 - 1) / (S / 64 * 64) == (nSamples + S - 1) / S) S = S / 64 * 64;
    }
    // the largest pass of the wanted shape that fits `room` bytes: fewer samples first, then (from one sample) fewer pixels
    auto shape = [&](size_t room, uint32_t &s, uint32_t &px) {
        s = S; px = pixPerPass;
        if ((size_t)px * (s * perItem + perPixel) <= room) return;
        const size_t sFit = room / px > perPixel ? (room / px - perPixel) / perItem : 0;
        if (sFit >= 1) s = (uint32_t)std::min<size_t>(s, sFit);
        else { s = 1; px = (uint32_t)std::min<size_t>(px, room / (perItem + perPixel) / 64 * 64); }
    };
    // as many contexts as wanted, but never more than there are passes (a call that is one pass runs it in one context at full size)
    uint32_t nPasses = 0;
    for (;; --nCtx) {
        uint32_t s, px;
        shape(limit / (size_t)nCtx, s, px);
        if (px > 0) nPasses = ((ds->nPix + px - 1) / px) * ((nSamples + s - 1) / s);
        if (nCtx > 1 && (px == 0 || nPasses < (uint32_t)nCtx)) continue;
        if (px == 0) return kz_fail(KZ_ERR_OOM, "64 (pixel, sample) items need %zu bytes of path state, the limit is %zu", (size_t)64 * (perItem + perPixel), limit);
        S = s; pixPerPass = px;
        break;
    }
    const size_t need = (size_t)pixPerPass * S;
    if (need >= (1ull << 32)) return kz_fail(KZ_ERR_UNSUPPORTED, "pass of %zu items (limit 2^32)", need);
    // Several passes in flight on internal streams when the call has at least two: the persistent traversal kernels of one pass
    // drain (fewer and fewer busy waves) while the other passes keep the machine full. The passes are independent except for the
    // film, whose read-modify-write kernel is chained with events in pass order.
    const bool multi = pipeline == 2 && nPasses >= 2 && nCtx >= 2;
    if (!multi) nCtx = 1;
    {   // contexts this call does not use are released when their memory is needed
        size_t keep = 0;
        for (int i = 0; i < nCtx; ++i) keep += std::max(ds->ctx[i].bytes(), need * perItem + pixPerPass * perPixel);
        for (int i = KZ_MAX_PASSES_IN_FLIGHT - 1; i >= nCtx; --i)
            if (ds->ctx[i].bytes() && keep + ds->ctx[i].bytes() > limit) { HIP_TRY(hipDeviceSynchronize()); ds->ctx[i].release(); } else keep += ds->ctx[i].bytes();
    }
    if (!ds->evCallA) { HIP_TRY(hipEventCreate(&ds->evCallA)); HIP_TRY(hipEventCreate(&ds->evCallB)); }
    if (beams && pixPerPass == ds->nPix) {                            // the shared lists of the whole pixel set
        if (!ds->evBeam) HIP_TRY(hipEventCreateWithFlags(&ds->evBeam, hipEventDisableTiming));
        if (ds->nPix > ds->beamCap) {
            HIP_TRY(hipDeviceSynchronize());
            if (ds->beamEntries) (void)hipFree(ds->beamEntries);
            if (ds->beamCount) (void)hipFree(ds->beamCount);
            ds->beamEntries = nullptr; ds->beamCount = nullptr; ds->beamCap = 0; ds->beamGen = 0;
            KZ_ALLOC(&ds->beamEntries, (size_t)ds->nPix * KZ_BEAM_CAP * sizeof(uint2));
            KZ_ALLOC(&ds->beamCount, (size_t)ds->nPix * sizeof(uint2));
            ds->beamCap = ds->nPix;
        }
    }
    if (multi) {
        // Different priorities put streams on different hardware queues whatever other streams the process has created (streams of one
        // priority share a small round-robin pool of queues and two of them may end up serialised on one): the pass streams cycle
        // through the priority levels the device has.
        int prLeast = 0, prGreatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prLeast, &prGreatest));
        const int mode = opts->tune.streamPriority > 0 ? opts->tune.streamPriority : 3;
        if (mode != ds->streamMode) {                                  // another policy than the streams were made with: make them again
            HIP_TRY(hipDeviceSynchronize());
            for (hipStream_t &st : ds->passStream) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
            ds->streamMode = mode;
        }
        const int nPr = std::max(1, prLeast - prGreatest + 1);
        for (int i = 0; i < nCtx; ++i) {
            if (!ds->evFilm[i]) HIP_TRY(hipEventCreateWithFlags(&ds->evFilm[i], hipEventDisableTiming));
            if (ds->passStream[i]) continue;
            const int pr = mode == 1 ? (prLeast + prGreatest) / 2 : mode == 2 ? ((i & 1) ? prGreatest : prLeast) : prLeast - (i % nPr);
            HIP_TRY(hipStreamCreateWithPriority(&ds->passStream[i], hipStreamNonBlocking, pr));
        }
        if (!ds->evFork) HIP_TRY(hipEventCreateWithFlags(&ds->evFork, hipEventDisableTiming));
    }
    ds->eventsUsed = 0;
    HIP_TRY(hipEventRecord(ds->evCallA, stream));
    if (multi) {
        HIP_TRY(hipEventRecord(ds->evFork, stream));
        for (int i = 0; i < nCtx; ++i) HIP_TRY(hipStreamWaitEvent(ds->passStream[i], ds->evFork, 0));
    }
    const int cols = P.width + 2 * P.border, rows = P.height + 2 * P.border;
    uint32_t pass = 0;
    for (uint32_t p0 = 0; p0 < ds->nPix; p0 += pixPerPass) {
        const uint32_t nPixPass = std::min(pixPerPass, ds->nPix - p0);
        const uint32_t *pixList = ds->pixList + p0;
        for (uint32_t s = s0; s < s1; s += S, ++pass) {
            const uint32_t Sp = std::min(S, s1 - s);
            const size_t items = (size_t)nPixPass * Sp;
            const int ci = multi ? (int)(pass % (uint32_t)nCtx) : 0;
            PassCtx &c = ds->ctx[ci];
            hipStream_t pst = multi ? ds->passStream[ci] : stream;
            if ((rc = ctxEnsure(c, need, pixPerPass, pipeline == 2, tapSums, beams && pixPerPass != ds->nPix, pst))) return rc;
            if (ds->eventsUsed == ds->events.size()) {
                EventPair ep; HIP_TRY(hipEventCreate(&ep.a)); HIP_TRY(hipEventCreate(&ep.b)); ds->events.push_back(ep);
            }
            EventPair &ep = ds->events[ds->eventsUsed++];
            const dim3 grid((unsigned)((items + KZ_BLOCK - 1) / KZ_BLOCK));
            float *sJx = c.samp, *sJy = c.samp + c.sampCap, *sR = c.samp + 2 * c.sampCap, *sG = c.samp + 3 * c.sampCap, *sB = c.samp + 4 * c.sampCap;
            HIP_TRY(hipEventRecord(ep.a, pst));
            if (pipeline == 2) { if ((rc = wfPass(scene, ds, c, pst, pixList, p0, nPixPass, s, Sp, (uint32_t)items, tune, beams))) return rc; }
            else {
#define KZ_MEGA(ST, EX) hipLaunchKernelGGL((kz_path_megakernel<ST, EX>), grid, dim3(KZ_BLOCK), 0, pst, P, ds->T, pixList, (uint32_t)items, Sp, s, \
                                           (const uint32_t *)nullptr, sJx, sJy, sR, sG, sB, ds->stats)
                if (ds->statsOn) { if (P.bsdfExt) KZ_MEGA(true, true); else KZ_MEGA(true, false); }
                else { if (P.bsdfExt) KZ_MEGA(false, true); else KZ_MEGA(false, false); }
#undef KZ_MEGA
            }
            HIP_TRY(hipEventRecord(ep.b, pst));
            HIP_TRY(hipGetLastError());
            const dim3 fgrid((cols + KZ_FILM_TILE - 1) / KZ_FILM_TILE, (rows + KZ_FILM_TILE - 1) / KZ_FILM_TILE);
            const int prev = (ci + nCtx - 1) % nCtx;
            if (tapSums) {
                // two kernels, every sample record read once. The tap sums do not depend on the film: only kz_film_apply waits for the film of the pass before.
#define KZ_FILM_TAPS(N) hipLaunchKernelGGL(kz_film_taps<N>, dim3((nPixPass + 63) / 64), dim3(64), 0, pst, P, ds->T.filter, pixList, nPixPass, Sp, sJx, sJy, sR, sG, sB, (float4 *)c.taps)
                switch (ftaps) { case 1: KZ_FILM_TAPS(1); break; case 2: KZ_FILM_TAPS(2); break; case 3: KZ_FILM_TAPS(3); break; case 4: KZ_FILM_TAPS(4); break; default: KZ_FILM_TAPS(5); break; }
#undef KZ_FILM_TAPS
                if (multi && pass > 0) HIP_TRY(hipStreamWaitEvent(pst, ds->evFilm[prev], 0));
                hipLaunchKernelGGL(kz_film_apply, fgrid, dim3(256), 0, pst, P, ds->pixIndex, (const float4 *)c.taps, p0, nPixPass, ds->film);
            } else {
                const int fr = KZ_FILM_TILE + ftaps - 1;
                const size_t perSample = (size_t)(3 + 2 * ftaps) * fr * fr * sizeof(float);
                const int fchunk = (int)std::max<size_t>(1, std::min<size_t>({(size_t)Sp, (size_t)8, (size_t)(64 * 1024) / perSample}));
                const size_t fshm = perSample * fchunk;
                if (multi && pass > 0) HIP_TRY(hipStreamWaitEvent(pst, ds->evFilm[prev], 0));
                hipLaunchKernelGGL(kz_film_gather, fgrid, dim3(256), fshm, pst, P, ds->T.filter, ds->pixIndex, p0, nPixPass, Sp, fchunk, sJx, sJy, sR, sG, sB, ds->film);
            }
            HIP_TRY(hipGetLastError());
            if (multi) HIP_TRY(hipEventRecord(ds->evFilm[ci], pst));
            if (pipeline == 2) { int rc_ = stageMark(c, pst, 4); if (rc_) return rc_; }
            ds->lastCtx = ci;
        }
    }
    if (multi)                                                         // join: everything after this call on `stream` sees the film
        for (int i = 0; i < nCtx && (uint32_t)i < pass; ++i) HIP_TRY(hipStreamWaitEvent(stream, ds->evFilm[i], 0));
    HIP_TRY(hipEventRecord(ds->evCallB, stream));
    ds->lastDual = multi;
    ds->lastInfo.passes = pass; ds->lastInfo.passesInFlight = (uint32_t)nCtx; ds->lastInfo.itemsPerPass = need; ds->lastInfo.sppPerPass = S;
    ds->lastInfo.pixels = ds->nPix; ds->lastInfo.stateBytes = ds->ctxBytes(); ds->lastInfo.pixelsPerPass = pixPerPass;
    return KZ_OK;
}

// Floats of the packed film rects of a tile list: tile t holds (h + 2b) x (w + 2b) x 4 floats - the tile with its filter apron.
static size_t packedFloats(const KzParams &P, const KzTile *tiles, uint32_t nTiles) {
    size_t n = 0;
    for (uint32_t t = 0; t < nTiles; ++t) n += (size_t)(tiles[t].w + 2 * P.border) * (size_t)(tiles[t].h + 2 * P.border) * 4;
    return n;
}
static int checkTiles(const KzParams &P, const KzTile *tiles, uint32_t nTiles) {
    if (!tiles && nTiles) return kz_fail(KZ_ERR_INVALID_ARG, "null tile list");
    for (uint32_t t = 0; t < nTiles; ++t) {
        const KzTile &tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.w <= 0 || tl.h <= 0 || tl.x0 + tl.w > P.width || tl.y0 + tl.h > P.height)
            return kz_fail(KZ_ERR_INVALID_ARG, "tile %u (%d,%d %dx%d) outside the %dx%d image", t, tl.x0, tl.y0, tl.w, tl.h, P.width, P.height);
    }
    return KZ_OK;
}

// the film rects of `tiles` of replica ds -> host `packed` (through a device-side pack and a pinned staging buffer: one D2H copy at link rate)
static int downloadTiles(KzScene *scene, KzDeviceState *ds, const KzTile *tiles, uint32_t nTiles, float *packed, size_t nFloats, hipStream_t stream) {
    const KzParams &P = scene->prm;
    int rc;
    if ((rc = checkTiles(P, tiles, nTiles))) return rc;
    const size_t need = packedFloats(P, tiles, nTiles);
    if (!packed || nFloats != need) return kz_fail(KZ_ERR_INVALID_ARG, "packed tile buffer must hold %zu floats (kz_tiles_packed_floats)", need);
    if (nTiles == 0) return KZ_OK;
    std::vector<KzTileRect> rects(nTiles);
    std::vector<uint32_t> prev;
    size_t off = 0; int maxRows = 0;
    for (uint32_t t = 0; t < nTiles; ++t) {
        rects[t] = KzTileRect{tiles[t].x0, tiles[t].y0, tiles[t].w, tiles[t].h, (uint32_t)off, (uint32_t)prev.size(), 0u};
        off += (size_t)(tiles[t].w + 2 * P.border) * (size_t)(tiles[t].h + 2 * P.border);
        maxRows = std::max(maxRows, tiles[t].h + 2 * P.border);
    }
    {   // earlier tiles whose rect (tile + apron) overlaps a tile's: a sweep over the tiles sorted by y keeps this near linear
        std::vector<uint32_t> byY(nTiles);
        for (uint32_t t = 0; t < nTiles; ++t) byY[t] = t;
        std::sort(byY.begin(), byY.end(), [&](uint32_t a, uint32_t b) { return tiles[a].y0 < tiles[b].y0; });
        std::vector<std::vector<uint32_t>> pv(nTiles);
        const int b2 = 2 * P.border;
        for (uint32_t i = 0; i < nTiles; ++i) {
            const KzTile &a = tiles[byY[i]];
            for (uint32_t j = i + 1; j < nTiles; ++j) {
                const KzTile &c = tiles[byY[j]];
                if (c.y0 >= a.y0 + a.h + b2) break;                       // sorted by y0: nothing further down overlaps a
                if (c.x0 < a.x0 + a.w + b2 && a.x0 < c.x0 + c.w + b2) {   // (y ranges overlap by the break test and the sort)
                    const uint32_t lo = std::min(byY[i], byY[j]), hi = std::max(byY[i], byY[j]);
                    pv[hi].push_back(lo);
                }
            }
        }
        for (uint32_t t = 0; t < nTiles; ++t) { rects[t].prevStart = (uint32_t)prev.size(); rects[t].prevCount = (uint32_t)pv[t].size(); prev.insert(prev.end(), pv[t].begin(), pv[t].end()); }
    }
    if (off >= (1ull << 32)) return kz_fail(KZ_ERR_UNSUPPORTED, "tile set of %zu film pixels (limit 2^32)", off);
    if (nTiles > ds->rectsCap) {
        if (ds->rectsDev) (void)hipFree(ds->rectsDev);
        ds->rectsDev = nullptr; ds->rectsCap = 0;
        KZ_ALLOC(&ds->rectsDev, (size_t)nTiles * sizeof(KzTileRect));
        ds->rectsCap = nTiles;
    }
    if (off > ds->packCap) {
        if (ds->packDev) (void)hipFree(ds->packDev);
        ds->packDev = nullptr; ds->packCap = 0;
        const size_t cap = off + off / 8;                              // (headroom: the next tile set of about this size reuses the buffers)
        KZ_ALLOC(&ds->packDev, cap * sizeof(float4));
        ds->packCap = cap;
    }
    if (off > ds->packHostCap) {
        if (ds->packHost) (void)hipHostFree(ds->packHost);
        ds->packHost = nullptr; ds->packHostCap = 0;
        const size_t cap = off + off / 8;
        if (hipHostMalloc((void **)&ds->packHost, cap * sizeof(float4), hipHostMallocDefault) == hipSuccess) ds->packHostCap = cap;
        else ds->packHost = nullptr;                                   // (no pinned memory to be had: the copy below goes to the caller's pageable buffer)
    }
    if (prev.size() + 1 > ds->prevCap) {
        if (ds->prevDev) (void)hipFree(ds->prevDev);
        ds->prevDev = nullptr; ds->prevCap = 0;
        KZ_ALLOC(&ds->prevDev, (prev.size() + 1) * sizeof(uint32_t));
        ds->prevCap = prev.size() + 1;
    }
    HIP_TRY(hipMemcpyAsync(ds->rectsDev, rects.data(), (size_t)nTiles * sizeof(KzTileRect), hipMemcpyHostToDevice, stream));
    if (!prev.empty()) HIP_TRY(hipMemcpyAsync(ds->prevDev, prev.data(), prev.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));                              // (the tables are host vectors of this call)
    hipLaunchKernelGGL(kz_film_pack, dim3((unsigned)maxRows, nTiles), dim3(128), 0, stream, (const float4 *)ds->film, P.width + 2 * P.border, (const KzTileRect *)ds->rectsDev,
                       (const uint32_t *)ds->prevDev, P.border, ds->packDev);
    HIP_TRY(hipGetLastError());
    if (ds->packHost) {
        HIP_TRY(hipMemcpyAsync(ds->packHost, ds->packDev, off * sizeof(float4), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        std::memcpy(packed, ds->packHost, off * sizeof(float4));
    } else {
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(packed, ds->packDev, off * sizeof(float4), hipMemcpyDeviceToHost));
    }
    return KZ_OK;
}

extern "C" {

int kz_render(KzScene *scene, const KzRenderOpts *opts) {
    if (!opts) return kz_fail(KZ_ERR_INVALID_ARG, "null opts");
    KzDeviceState *ds; int rc;
    // opts->device addresses a replica by HIP device index; a scene resident on ONE device is addressed by any zero-initialised opts
    if ((rc = findReplica(scene, opts->device, &ds))) {
        KzReplicaSet *rs = scene ? replicaSet(scene) : nullptr;
        bool single = false;
        if (rs) { std::lock_guard<std::mutex> g(rs->m); single = rs->v.size() == 1 && opts->device == 0; }
        if (!single || (rc = findReplica(scene, -1, &ds))) return rc;
    }
    return renderOn(scene, ds, opts);
}

int kz_tiles_packed_floats(const KzScene *scene, const KzTile *tiles, uint32_t nTiles, size_t *nFloats) {
    if (!scene || !nFloats) return kz_fail(KZ_ERR_INVALID_ARG, "null argument");
    int rc;
    if ((rc = checkTiles(scene->prm, tiles, nTiles))) return rc;
    *nFloats = packedFloats(scene->prm, tiles, nTiles);
    return KZ_OK;
}

int kz_film_download_tiles(KzScene *scene, int device, const KzTile *tiles, uint32_t nTiles, float *packed, size_t nFloats) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    return downloadTiles(scene, ds, tiles, nTiles, packed, nFloats, ds->lastStream);
}

int kz_render_tiles(KzScene *scene, const KzRenderOpts *opts, const KzTile *tiles, uint32_t nTiles, int device, float *film, size_t nFloats) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    const size_t full = ds->filmPixels * 4;
    size_t packed = 0;
    if (film) {
        if ((rc = checkTiles(scene->prm, tiles, nTiles))) return rc;
        packed = packedFloats(scene->prm, tiles, nTiles);
        if (nFloats != full && !(nTiles && nFloats == packed))
            return kz_fail(KZ_ERR_INVALID_ARG, "film buffer must hold the tiles' packed rects (%zu floats, kz_tiles_packed_floats) or the whole film (%zu floats)", packed, full);
    }
    KzRenderOpts o{};
    if (opts) o = *opts;
    o.tiles = tiles; o.nTiles = nTiles; o.device = device;
    if ((rc = renderOn(scene, ds, &o))) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)o.stream));
    if (film && nFloats == full && !(nTiles && packed == full)) HIP_TRY(hipMemcpy(film, ds->film, nFloats * sizeof(float), hipMemcpyDeviceToHost));
    else if (film) return downloadTiles(scene, ds, tiles, nTiles, film, nFloats, (hipStream_t)o.stream);
    return KZ_OK;
}

int kz_deal_tiles(int32_t width, int32_t height, int32_t tileSize, uint32_t nParts, uint32_t part, KzTile *out, uint32_t cap, uint32_t *count) {
    if (tileSize == 0) tileSize = 64;
    if (width <= 0 || height <= 0 || tileSize < 32 || tileSize % 32 || nParts == 0 || part >= nParts || !count)
        return kz_fail(KZ_ERR_INVALID_ARG, "kz_deal_tiles: bad argument (tile size must be a positive multiple of 32, part < nParts)");
    std::vector<KzTile> tiles;
    for (int y = 0; y < height; y += tileSize)
        for (int x = 0; x < width; x += tileSize) tiles.push_back(KzTile{x, y, std::min(tileSize, width - x), std::min(tileSize, height - y)});
    // largest first (stable: row-major order among equals), each to the part with the least area so far (ties: the lower part)
    std::vector<uint32_t> order(tiles.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return (int64_t)tiles[a].w * tiles[a].h > (int64_t)tiles[b].w * tiles[b].h; });
    std::vector<int64_t> area(nParts, 0);
    std::vector<uint32_t> mine;
    for (uint32_t i : order) {
        uint32_t best = 0;
        for (uint32_t p = 1; p < nParts; ++p) if (area[p] < area[best]) best = p;
        area[best] += (int64_t)tiles[i].w * tiles[i].h;
        if (best == part) mine.push_back(i);
    }
    std::sort(mine.begin(), mine.end());                                // back to row-major order within the part
    *count = (uint32_t)mine.size();
    if (mine.size() > cap || (!out && !mine.empty())) return kz_fail(KZ_ERR_INVALID_ARG, "kz_deal_tiles: %zu tiles, room for %u", mine.size(), cap);
    for (size_t i = 0; i < mine.size(); ++i) out[i] = tiles[mine[i]];
    return KZ_OK;
}

int kz_film_merge(float *dst, const float *src, size_t nFloats) {
    if (!dst || !src) return kz_fail(KZ_ERR_INVALID_ARG, "null film");
    for (size_t i = 0; i < nFloats; ++i) dst[i] += src[i];
    return KZ_OK;
}

// ImageBlock::put(ImageBlock&) (block.cpp:87-96) for a LIST of blocks: the packed rects of `tiles` are added to the film in list order.
// Rows of the film are cut into bands, one host thread per band (disjoint destinations: no lock, and every film texel still receives
// its rects in list order, so the result does not depend on the number of threads).
int kz_film_merge_tiles(float *film, int32_t width, int32_t height, int32_t border, const KzTile *tiles, uint32_t nTiles, const float *packed, size_t nFloats, int32_t nThreads) {
    if (!film || (nTiles && (!tiles || !packed)) || width <= 0 || height <= 0 || border < 0) return kz_fail(KZ_ERR_INVALID_ARG, "kz_film_merge_tiles: null or bad argument");
    const int cols = width + 2 * border, rows = height + 2 * border;
    std::vector<size_t> offs(nTiles);
    size_t off = 0;
    for (uint32_t t = 0; t < nTiles; ++t) {
        const KzTile &tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.w <= 0 || tl.h <= 0 || tl.x0 + tl.w > width || tl.y0 + tl.h > height) return kz_fail(KZ_ERR_INVALID_ARG, "tile %u outside the %dx%d image", t, width, height);
        offs[t] = off; off += (size_t)(tl.w + 2 * border) * (size_t)(tl.h + 2 * border) * 4;
    }
    if (off != nFloats) return kz_fail(KZ_ERR_INVALID_ARG, "packed buffer holds %zu floats, the tiles need %zu", nFloats, off);
    int nt = nThreads > 0 ? nThreads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    nt = std::max(1, std::min(nt, rows / 8 + 1));
    auto band = [&](int r0, int r1) {
        for (uint32_t t = 0; t < nTiles; ++t) {
            const KzTile &tl = tiles[t];
            const int rw = tl.w + 2 * border, y0 = std::max(tl.y0, r0), y1 = std::min(tl.y0 + tl.h + 2 * border, r1);
            for (int y = y0; y < y1; ++y) {
                float *d = film + ((size_t)y * cols + tl.x0) * 4;
                const float *s = packed + offs[t] + (size_t)(y - tl.y0) * rw * 4;
                for (int i = 0; i < rw * 4; ++i) d[i] += s[i];
            }
        }
    };
    if (nt == 1) { band(0, rows); return KZ_OK; }
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(band, (int)((int64_t)rows * i / nt), (int)((int64_t)rows * (i + 1) / nt));
    for (auto &t : th) t.join();
    return KZ_OK;
}

int kz_render_multi(KzScene *scene, const KzRenderOpts *opts, const int32_t *devices, uint32_t nDevices, int32_t tileSize, float *film, size_t nFloats,
                    float *deviceMs) {
    if (!scene || !devices || nDevices == 0 || !film) return kz_fail(KZ_ERR_INVALID_ARG, "kz_render_multi: null argument");
    const KzParams &P = scene->prm;
    const size_t filmFloats = (size_t)(P.width + 2 * P.border) * (size_t)(P.height + 2 * P.border) * 4;
    if (nFloats != filmFloats) return kz_fail(KZ_ERR_INVALID_ARG, "film buffer must hold %zu floats", filmFloats);
    for (uint32_t i = 0; i < nDevices; ++i) for (uint32_t j = 0; j < i; ++j) if (devices[i] == devices[j]) return kz_fail(KZ_ERR_INVALID_ARG, "device %d listed twice", devices[i]);
    // the frame's tiles in row-major order: the unit of dealing AND of the merge (a film texel receives the rects that reach it in TILE order,
    // whichever device rendered them: the result is the same for static and for dynamic dealing, and from run to run)
    uint32_t nAll = 0;
    (void)kz_deal_tiles(P.width, P.height, tileSize, 1, 0, nullptr, 0, &nAll);
    std::vector<KzTile> all(nAll);
    int rc = nAll ? kz_deal_tiles(P.width, P.height, tileSize, 1, 0, all.data(), nAll, &nAll) : KZ_OK;
    if (rc) return rc;
    // replicas come up BEFORE the clocks start (deviceMs is render + gather; a first call pays the upload outside it)
    for (uint32_t i = 0; i < nDevices; ++i) if ((rc = kz_scene_upload(scene, devices[i]))) { const std::string why = kz_last_error(); return kz_fail(rc, "device %d: %s", devices[i], why.c_str()); }
    const bool dynamic = opts && opts->tileDealing == 1;
    struct Job { std::vector<KzTile> tiles; std::vector<float> packed; int rc = KZ_OK; std::string err; float ms = 0.f; };
    std::vector<Job> jobs(nDevices);
    if (!dynamic) {
        for (uint32_t i = 0; i < nDevices; ++i) {
            uint32_t n = 0;
            (void)kz_deal_tiles(P.width, P.height, tileSize, nDevices, i, nullptr, 0, &n);
            jobs[i].tiles.resize(n);
            if (n && (rc = kz_deal_tiles(P.width, P.height, tileSize, nDevices, i, jobs[i].tiles.data(), n, &n))) return rc;
        }
    }
    // dynamic dealing (the reference's BlockGenerator::next under a mutex, block.cpp:117-148): the workers pull batches of tiles - about two
    // passes' worth of (pixel, sample) items each - from one counter until the frame is dealt; a slow device simply takes fewer batches
    std::atomic<uint32_t> nextTile{0};
    uint32_t s0 = opts ? opts->sampleBegin : 0, s1 = opts ? opts->sampleEnd : 0;
    if (s0 == 0 && s1 == 0) s1 = P.sampleCount;
    const uint64_t itemsPerTile = (uint64_t)(tileSize ? tileSize : 64) * (tileSize ? tileSize : 64) * std::max<uint32_t>(1, s1 - s0);
    const uint64_t passItems = opts && opts->passItems ? opts->passItems : (1ull << 27);
    const uint32_t batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((2 * passItems + itemsPerTile - 1) / itemsPerTile, std::max<uint32_t>(1, nAll / (4 * nDevices))));
    // one host thread per device (renderer.cpp:94-127 runs one TBB task per block; here a task is a GPU's share of the tiles)
    std::vector<std::thread> threads;
    for (uint32_t i = 0; i < nDevices; ++i) {
        threads.emplace_back([&, i]() {
            Job &j = jobs[i];
            const auto t0 = std::chrono::steady_clock::now();
            KzRenderOpts o{};
            if (opts) o = *opts;
            o.stream = nullptr; o.accumulate = 0;
            if (dynamic) {
                for (;;) {
                    const uint32_t b = nextTile.fetch_add(batch);
                    if (b >= nAll) break;
                    const uint32_t e = std::min(nAll, b + batch);
                    j.rc = kz_render_tiles(scene, &o, all.data() + b, e - b, devices[i], nullptr, 0);
                    if (j.rc) break;
                    j.tiles.insert(j.tiles.end(), all.begin() + b, all.begin() + e);
                    o.accumulate = 1;                                    // the device film collects the batches
                }
            } else if (!j.tiles.empty()) j.rc = kz_render_tiles(scene, &o, j.tiles.data(), (uint32_t)j.tiles.size(), devices[i], nullptr, 0);
            if (!j.rc && !j.tiles.empty()) {
                j.packed.resize(packedFloats(P, j.tiles.data(), (uint32_t)j.tiles.size()));
                j.rc = kz_film_download_tiles(scene, devices[i], j.tiles.data(), (uint32_t)j.tiles.size(), j.packed.data(), j.packed.size());
            }
            if (j.rc) j.err = kz_last_error();                             // the message is thread-local: carry it to the caller's thread
            j.ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    for (auto &t : threads) t.join();
    for (uint32_t i = 0; i < nDevices; ++i) {
        if (deviceMs) deviceMs[i] = jobs[i].ms;
        if (jobs[i].rc) return kz_fail(jobs[i].rc, "device %d: %s", devices[i], jobs[i].err.c_str());
    }
    // ImageBlock::put(ImageBlock&) (block.cpp:87-96) in TILE order: a table (tile -> device, offset in that device's packed buffer), then one merge
    // over row bands; the rects are gathered into one list so that kz_film_merge_tiles sees them in tile order
    struct Rect { KzTile t; const float *src; };
    std::vector<Rect> tab;
    for (uint32_t i = 0; i < nDevices; ++i) {
        size_t off = 0;
        for (const KzTile &t : jobs[i].tiles) { tab.push_back(Rect{t, jobs[i].packed.data() + off}); off += (size_t)(t.w + 2 * P.border) * (size_t)(t.h + 2 * P.border) * 4; }
    }
    std::sort(tab.begin(), tab.end(), [](const Rect &a, const Rect &b) { return a.t.y0 != b.t.y0 ? a.t.y0 < b.t.y0 : a.t.x0 < b.t.x0; });
    std::memset(film, 0, filmFloats * sizeof(float));
    const int cols = P.width + 2 * P.border, rows = P.height + 2 * P.border, border = P.border;
    const int nt = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto band = [&](int r0, int r1) {
        for (const Rect &e : tab) {
            const int rw = e.t.w + 2 * border, y0 = std::max(e.t.y0, r0), y1 = std::min(e.t.y0 + e.t.h + 2 * border, r1);
            for (int y = y0; y < y1; ++y) {
                float *d = film + ((size_t)y * cols + e.t.x0) * 4;
                const float *sp = e.src + (size_t)(y - e.t.y0) * rw * 4;
                for (int k = 0; k < rw * 4; ++k) d[k] += sp[k];
            }
        }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < nt; ++i) th.emplace_back(band, (int)((int64_t)rows * i / nt), (int)((int64_t)rows * (i + 1) / nt));
    for (auto &t : th) t.join();
    return KZ_OK;
}

// Device time of the stages of the LAST pass of the last kz_render (wavefront pipeline), from hipEvents on the launch stream:
// out[0] generate, [1] closest-hit traversal of the BOUNCE rays (the kz_wf_trace<0> launches), [2] shade, [3] shadow traversal, [4] film,
// [5] camera rays (kz_wf_beam when the lists are rebuilt, kz_wf_trace_list, kz_wf_trace_packet, the first-hit walk-through).
int kz_last_stage_ms(KzScene *scene, float *out6) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!out6) return kz_fail(KZ_ERR_INVALID_ARG, "null out");
    for (int i = 0; i < 6; ++i) out6[i] = 0.f;
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    const PassCtx &c = ds->ctx[ds->lastCtx];
    for (size_t i = 1; i < c.stageUsed; ++i) {
        float t = 0; HIP_TRY(hipEventElapsedTime(&t, c.stageEv[i - 1], c.stageEv[i]));
        const int k = c.stageKind[i];
        if (k >= 0 && k < 6) out6[k] += t;
    }
    return KZ_OK;
}

int kz_last_pass_info(KzScene *scene, KzPassInfo *out) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!out) return kz_fail(KZ_ERR_INVALID_ARG, "null out");
    *out = ds->lastInfo;
    return KZ_OK;
}

int kz_sync_on(KzScene *scene, int device) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    return KZ_OK;
}
int kz_sync(KzScene *scene) { return kz_sync_on(scene, -1); }

int kz_last_kernel_ms(KzScene *scene, float *ms) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!ms) return kz_fail(KZ_ERR_INVALID_ARG, "null ms");
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    if (ds->lastDual && ds->lastInfo.passes) {          // passes overlap: the per-pass figure is the span of the call over its passes (film included)
        float t = 0; HIP_TRY(hipEventElapsedTime(&t, ds->evCallA, ds->evCallB));
        *ms = t / (float)ds->lastInfo.passes;
        return KZ_OK;
    }
    double tot = 0;
    for (size_t i = 0; i < ds->eventsUsed; ++i) { float t = 0; HIP_TRY(hipEventElapsedTime(&t, ds->events[i].a, ds->events[i].b)); tot += t; }
    *ms = ds->eventsUsed ? (float)(tot / ds->eventsUsed) : 0.f;
    return KZ_OK;
}

int kz_film_clear_on(KzScene *scene, int device, void *stream) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    HIP_TRY(hipMemsetAsync(ds->film, 0, ds->filmPixels * sizeof(float4), (hipStream_t)stream));
    return KZ_OK;
}
int kz_film_clear(KzScene *scene, void *stream) { return kz_film_clear_on(scene, -1, stream); }

int kz_film_download_on(KzScene *scene, int device, float *film, size_t nFloats) {
    KzDeviceState *ds; int rc;
    if ((rc = findReplica(scene, device, &ds))) return rc;
    if (!film || nFloats != ds->filmPixels * 4) return kz_fail(KZ_ERR_INVALID_ARG, "film buffer must hold %zu floats", ds->filmPixels * 4);
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    HIP_TRY(hipMemcpy(film, ds->film, nFloats * sizeof(float), hipMemcpyDeviceToHost));
    return KZ_OK;
}
int kz_film_download(KzScene *scene, float *film, size_t nFloats) { return kz_film_download_on(scene, -1, film, nFloats); }

// ImageBlock::toBitmap (block.cpp:39-45) + Bitmap::savePNG's tone map (bitmap.cpp:45-52): the film is resolved to the 8-bit
// sRGB raster on the device, so the host link carries 3 bytes per pixel instead of the 16-byte film texel.
int kz_film_to_srgb8(KzScene *scene, uint8_t *rgb8, size_t nBytes) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    const KzParams &P = scene->prm;
    const size_t need = (size_t)P.width * (size_t)P.height * 3;
    if (!rgb8 || nBytes != need) return kz_fail(KZ_ERR_INVALID_ARG, "rgb8 buffer must hold %zu bytes", need);
    if (!ds->srgb) KZ_ALLOC(&ds->srgb, need);                          // staging raster kept with the replica
    const uint32_t n = (uint32_t)(P.width * P.height);
    hipLaunchKernelGGL(kz_film_srgb8, dim3((n + 255) / 256), dim3(256), 0, ds->lastStream, ds->film, P.width, P.height, P.border, ds->srgb);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    HIP_TRY(hipMemcpy(rgb8, ds->srgb, need, hipMemcpyDeviceToHost));
    return KZ_OK;
}

int kz_set_stats(KzScene *scene, int enable) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    ds->statsOn = enable != 0;
    return KZ_OK;
}

int kz_get_stats(KzScene *scene, KzStats *out, int reset) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (!out) return kz_fail(KZ_ERR_INVALID_ARG, "null out");
    HIP_TRY(hipStreamSynchronize(ds->lastStream));
    unsigned long long h[8];
    HIP_TRY(hipMemcpy(h, ds->stats, sizeof h, hipMemcpyDeviceToHost));
    { unsigned long long b[3]; HIP_TRY(hipMemcpy(b, ds->stats + 24, sizeof b, hipMemcpyDeviceToHost)); out->beamPixels = b[0]; out->beamListEntries = b[1]; out->beamCompletePixels = b[2];
      if (reset) HIP_TRY(hipMemset(ds->stats + 24, 0, sizeof b)); }
    out->samples = h[0]; out->rays = h[1]; out->nodeVisits = h[2]; out->triTests = h[3]; out->shadedHits = h[4]; out->lightSamples = h[5]; out->droppedSamples = h[6];
#ifdef KZ_LANESTAT
    {
        unsigned long long ls[16];
        HIP_TRY(hipMemcpy(ls, ds->stats + 8, sizeof ls, hipMemcpyDeviceToHost));
        for (int m = 0; m < 2; ++m) {
            const unsigned long long *a = ls + 8 * m;
            std::fprintf(stderr, "lanestat %s: node iters %llu  active/iter %.2f  inner/iter %.2f | leaf phases %llu  lanes/phase %.2f  tri iters/phase %.2f | refills %llu  lanes/refill %.2f\n",
                         m ? "shadow " : "closest", a[0], a[0] ? (double)a[1] / a[0] : 0.0, a[0] ? (double)a[2] / a[0] : 0.0, a[3], a[3] ? (double)a[4] / a[3] : 0.0,
                         a[3] ? (double)a[7] / a[3] : 0.0, a[5], a[5] ? (double)a[6] / a[5] : 0.0);
        }
        if (reset) HIP_TRY(hipMemset(ds->stats + 8, 0, sizeof ls));
    }
#endif
#ifdef KZ_TRACESTAT
    {
        unsigned long long ts[10];
        HIP_TRY(hipMemcpy(ts, ds->stats + 8, sizeof ts, hipMemcpyDeviceToHost));
        for (int m = 0; m < 2; ++m) {
            const unsigned long long *a = ts + 4 * m; const double tot = (double)(a[0] + a[1] + a[2] + a[3]);
            std::fprintf(stderr, "tracestat %s: refill %.2f %%  node phase %.2f %%  leaf phase (+ pop / finish) %.2f %% (triangle loop alone %.2f %%)  loop %.2f %%\n", m ? "shadow " : "closest",
                         tot > 0 ? 100.0 * a[0] / tot : 0.0, tot > 0 ? 100.0 * a[1] / tot : 0.0, tot > 0 ? 100.0 * a[2] / tot : 0.0, tot > 0 ? 100.0 * ts[8 + m] / tot : 0.0, tot > 0 ? 100.0 * a[3] / tot : 0.0);
        }
        if (reset) HIP_TRY(hipMemset(ds->stats + 8, 0, sizeof ts));
    }
#endif
#ifdef KZ_SHADESTAT
    {
        unsigned long long ss[14]; double tot = 0;
        HIP_TRY(hipMemcpy(ss, ds->stats + 8, sizeof ss, hipMemcpyDeviceToHost));
        static const char *nm[14] = {"passA: rest (emitter, classification, roulette)", "compact+barrier", "B:loads+setup", "B:light draws+sample", "B:eval+pdf+shadow stores", "B:bsdf draws", "B:bsdf sample", "B:next-ray stores",
                                     "barrier+staging+flush", "loop head", "passA: queue+hit+ray loads", "passA: shading record arrives", "passA: postIntersect", "-"};
        for (int k = 0; k < 14; ++k) tot += (double)ss[k];
        for (int k = 0; k < 14; ++k) std::fprintf(stderr, "shadestat %-48s %6.2f %%\n", nm[k], tot > 0 ? 100.0 * (double)ss[k] / tot : 0.0);
        if (reset) HIP_TRY(hipMemset(ds->stats + 8, 0, sizeof ss));
    }
#endif
    if (reset) HIP_TRY(hipMemset(ds->stats, 0, sizeof h));
    return KZ_OK;
}

int kz_trace_rays(KzScene *scene, uint32_t n, const float *o, const float *d, const float *tmin, const float *tmax, KzHit *hits) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!o || !d || !tmin || !tmax || !hits) return kz_fail(KZ_ERR_INVALID_ARG, "null ray buffer");
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
int kz_render_samples(KzScene *scene, uint32_t n, const int32_t *pxy, const uint32_t *idx, float *out) {
    KzDeviceState *ds; int rc;
    if ((rc = requireDevice(scene, &ds))) return rc;
    if (n == 0) return KZ_OK;
    if (!pxy || !idx || !out) return kz_fail(KZ_ERR_INVALID_ARG, "null buffer");
    const KzParams &P = scene->prm;
    std::vector<uint32_t> pl(n);
    for (uint32_t i = 0; i < n; ++i) {
        if (pxy[2 * i] < 0 || pxy[2 * i] >= P.width || pxy[2 * i + 1] < 0 || pxy[2 * i + 1] >= P.height || idx[i] >= P.sampleCount)
            return kz_fail(KZ_ERR_INVALID_ARG, "sample %u: pixel (%d,%d) index %u out of range", i, pxy[2 * i], pxy[2 * i + 1], idx[i]);
        pl[i] = (uint32_t)pxy[2 * i] | ((uint32_t)pxy[2 * i + 1] << 16);
    }
    DevMem dP, dI, dOut;
    KZ_ALLOC(&dP.p, (size_t)n * 4); KZ_ALLOC(&dI.p, (size_t)n * 4); KZ_ALLOC(&dOut.p, (size_t)n * 20);
    float *dO = dOut.as<float>();
    HIP_TRY(hipMemcpy(dP.p, pl.data(), (size_t)n * 4, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(dI.p, idx, (size_t)n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((kz_path_megakernel<false, true>), dim3((n + KZ_BLOCK - 1) / KZ_BLOCK), dim3(KZ_BLOCK), 0, 0, P, ds->T, dP.as<uint32_t>(), n, 1u, 0u, dI.as<uint32_t>(),
                       dO, dO + n, dO + 2 * (size_t)n, dO + 3 * (size_t)n, dO + 4 * (size_t)n, ds->stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    std::vector<float> h((size_t)n * 5);
    HIP_TRY(hipMemcpy(h.data(), dO, (size_t)n * 20, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < n; ++i) {
        out[5 * i] = (float)pxy[2 * i] + h[i]; out[5 * i + 1] = (float)pxy[2 * i + 1] + h[n + i];
        out[5 * i + 2] = h[2 * (size_t)n + i]; out[5 * i + 3] = h[3 * (size_t)n + i]; out[5 * i + 4] = h[4 * (size_t)n + i];
    }
    return KZ_OK;
}

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
