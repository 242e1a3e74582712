// kz_film.hip - the film: deterministic ImageBlock::put (block.cpp:56-96) as HIP kernels for gfx950, the packing / download of tile rects
// (the device half of the multi-GPU gather), the 8-bit sRGB resolve, and the kz_film_* entry points.
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

// kz_film_taps with TWO lanes per pixel (the default since round 5; KzTuning.filmGather = 3 selects the one-lane kernel above; profiles/r05h_film_taps): the 25 (rgb w, w)
// accumulators of a pixel hold kz_film_taps<5> at 159 VGPRs and 3 waves per SIMD, and its staging area - 8 samples of 64 pixels - lets it read only 32-B pieces of the
// sample rows: every 64-B HBM sector is fetched twice (42 GB per 2^30-item pass for 21 GB of records, L2 hit 1 %: profiles/r04z_round4). Here a wave is 32 pixels: lanes 0-31 carry the tap ROWS 0 .. ceil(TAPS / 2) - 1 of their pixel, lanes 32-63 the
// remaining rows of the same 32 pixels - 15 and 10 accumulators instead of 25. Every accumulator still receives the same products in the same (sample) order, so the
// tap sums are kz_film_taps' bit for bit; what is paid is the per-sample set-up (validity, footprint, the x weights) twice. With half the pixels per wave the same
// 10.4 KB stage 16 samples a round: 64-B pieces, whole sectors. Film stage of a 2^30-item pass of C4 12.6 -> 10.5 ms, C3 (2^27 items) 1.98 -> 1.64 ms; occupancy alone
// (8-sample rounds, two lanes) changed nothing: the one-lane kernel was bound by its doubled fetch.
#ifndef KZ_TAPS2_CHUNK
#define KZ_TAPS2_CHUNK 16                    // samples per staging round: 64-B pieces of every sample row = whole HBM sectors (the 32-B pieces of kz_film_taps fetch every sector twice)
#endif
template <int TAPS>
__global__ __launch_bounds__(64) void kz_film_taps2(KzParams P, const float *__restrict__ filter, const uint32_t *__restrict__ pixList, uint32_t nPix, uint32_t S,
                                                    const float *__restrict__ inJx, const float *__restrict__ inJy, const float *__restrict__ inR,
                                                    const float *__restrict__ inG, const float *__restrict__ inB, float4 *__restrict__ tapSums) {
    constexpr int R0 = (TAPS + 1) / 2;                                 // tap rows of the first half (the second half carries TAPS - R0 <= R0 rows)
    __shared__ float s_filter[KZ_FILTER_RESOLUTION + 1];
    __shared__ float s_in[5][KZ_TAPS2_CHUNK][33];
    const int lane = threadIdx.x, pixLane = lane & 31, half = lane >> 5;
    if (lane <= KZ_FILTER_RESOLUTION) s_filter[lane] = filter[lane];
    const uint32_t pl0 = blockIdx.x * 32u, pl = pl0 + (uint32_t)pixLane;
    const bool havePixel = pl < nPix;
    const uint32_t pxy = havePixel ? pixList[pl] : 0u;
    const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
    const int bx0 = px & ~31, by0 = py & ~31;
    const float r = P.filterRadius, lf = P.lookupFactor;
    const int ty0 = half ? R0 : 0, nRowsMine = half ? TAPS - R0 : R0;
    float xb[TAPS], yb[R0];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) xb[t] = (float)(px + P.border - P.tapLo - t - bx0);
#pragma unroll
    for (int t = 0; t < R0; ++t) yb[t] = (float)(py + P.border - P.tapLo - (ty0 + t) - by0);
    float4 acc[R0 * TAPS];
#pragma unroll
    for (int i = 0; i < R0 * TAPS; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float *const in[5] = {inJx, inJy, inR, inG, inB};
    const uint32_t nRows = min(32u, nPix - min(nPix, pl0));
    for (uint32_t c0 = 0; c0 < S; c0 += KZ_TAPS2_CHUNK) {
        const uint32_t n = min((uint32_t)KZ_TAPS2_CHUNK, S - c0);
        __syncthreads();
        for (uint32_t i = lane; i < nRows * KZ_TAPS2_CHUNK; i += 64u) {
            const uint32_t p = i / KZ_TAPS2_CHUNK, k = i % KZ_TAPS2_CHUNK;
            if (k < n) {
                const size_t gi = (size_t)(pl0 + p) * S + c0 + k;
#pragma unroll
                for (int a = 0; a < 5; ++a) s_in[a][k][p] = in[a][gi];
            }
        }
        __syncthreads();
        if (havePixel) {
            for (uint32_t k = 0; k < n; ++k) {
                const float jx = s_in[0][k][pixLane], jy = s_in[1][k][pixLane];
                float cr = s_in[2][k][pixLane], cg = s_in[3][k][pixLane], cb = s_in[4][k][pixLane];
                const bool valid = cr >= 0.f && cg >= 0.f && cb >= 0.f && isfinite(cr) && isfinite(cg) && isfinite(cb);
                if (!valid) continue;
                const float posx = ((float)px + jx) - 0.5f - (float)(bx0 - P.border), posy = ((float)py + jy) - 0.5f - (float)(by0 - P.border);
                const float lox = ceilf(posx - r), hix = floorf(posx + r), loy = ceilf(posy - r), hiy = floorf(posy + r);
                float wx[TAPS], wy[R0];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) wx[t] = !(xb[t] < lox || xb[t] > hix) ? s_filter[(int)(fabsf(xb[t] - posx) * lf)] : 0.f;
#pragma unroll
                for (int t = 0; t < R0; ++t) wy[t] = (t < nRowsMine && !(yb[t] < loy || yb[t] > hiy)) ? s_filter[(int)(fabsf(yb[t] - posy) * lf)] : 0.f;
#pragma unroll
                for (int ty = 0; ty < R0; ++ty)
#pragma unroll
                    for (int tx = 0; tx < TAPS; ++tx) {
                        float4 &a = acc[ty * TAPS + tx];
                        a.x += cr * wx[tx] * wy[ty]; a.y += cg * wx[tx] * wy[ty]; a.z += cb * wx[tx] * wy[ty]; a.w += 1.0f * wx[tx] * wy[ty];
                    }
            }
        }
    }
    if (havePixel) {
#pragma unroll
        for (int ty = 0; ty < R0; ++ty)
            if (ty < nRowsMine) {
#pragma unroll
                for (int tx = 0; tx < TAPS; ++tx) tapSums[(size_t)((ty0 + ty) * TAPS + tx) * nPix + pl] = acc[ty * TAPS + tx];
            }
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
__global__ __launch_bounds__(128) void kz_film_pack(const float4 *__restrict__ film, int cols, const KzTileRect *__restrict__ rects, const uint32_t *__restrict__ prev,
                                                    int border, float4 *__restrict__ out, uint32_t nTiles) {
    const uint32_t tile = blockIdx.y + blockIdx.z * 65535u;             // (a grid's y extent ends at 65535: longer tile lists continue in z)
    if (tile >= nTiles) return;
    const KzTileRect r = rects[tile];
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
        const float t = v <= 0.0031308f ? 12.92f * v : (1.0f + 0.055f) * kzPow(v, 1.0f / 2.4f) - 0.055f;
        const float s = 255.f * t;
        out[3 * (size_t)i + k] = (uint8_t)(s < 0.f ? 0.f : (s > 255.f ? 255.f : s));
    }
}

// once per replica (kz_scene_upload): the staged gather kernel may ask for more than the default 64 KB of dynamic LDS
int kzFilmInit() {
    HIP_TRY(hipFuncSetAttribute((const void *)kz_film_gather, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    return KZ_OK;
}

// The film stage of one pass (called by renderOn, kz_render.hip). With tap sums (filters of <= KZ_TAPS_MAX taps per axis): two kernels, every sample record
// read once; the tap sums do not depend on the film, only kz_film_apply waits for the film of the pass before (`waitFilm`).
int kzFilmStage(KzScene *scene, KzDeviceState *ds, PassCtx &c, hipStream_t pst, const uint32_t *pixList, uint32_t p0, uint32_t nPixPass, uint32_t Sp, bool tapSums, hipEvent_t waitFilm, bool twoLanes) {
    const KzParams &P = scene->prm;
    const int cols = P.width + 2 * P.border, rows = P.height + 2 * P.border, ftaps = P.tapHi - P.tapLo + 1;
    float *sJx = c.plane[0], *sJy = c.plane[1], *sR = c.plane[2], *sG = c.plane[3], *sB = c.plane[4];
    const dim3 fgrid((cols + KZ_FILM_TILE - 1) / KZ_FILM_TILE, (rows + KZ_FILM_TILE - 1) / KZ_FILM_TILE);
    if (tapSums) {
#define KZ_FILM_TAPS(N) hipLaunchKernelGGL(kz_film_taps<N>, dim3((nPixPass + 63) / 64), dim3(64), 0, pst, P, ds->T.filter, pixList, nPixPass, Sp, sJx, sJy, sR, sG, sB, (float4 *)c.taps)
#define KZ_FILM_TAPS2(N) hipLaunchKernelGGL(kz_film_taps2<N>, dim3((nPixPass + 31) / 32), dim3(64), 0, pst, P, ds->T.filter, pixList, nPixPass, Sp, sJx, sJy, sR, sG, sB, (float4 *)c.taps)
        if (twoLanes) switch (ftaps) { case 1: KZ_FILM_TAPS2(1); break; case 2: KZ_FILM_TAPS2(2); break; case 3: KZ_FILM_TAPS2(3); break; case 4: KZ_FILM_TAPS2(4); break; default: KZ_FILM_TAPS2(5); break; }
        else switch (ftaps) { case 1: KZ_FILM_TAPS(1); break; case 2: KZ_FILM_TAPS(2); break; case 3: KZ_FILM_TAPS(3); break; case 4: KZ_FILM_TAPS(4); break; default: KZ_FILM_TAPS(5); break; }
#undef KZ_FILM_TAPS2
#undef KZ_FILM_TAPS
        if (waitFilm) HIP_TRY(hipStreamWaitEvent(pst, waitFilm, 0));
        hipLaunchKernelGGL(kz_film_apply, fgrid, dim3(256), 0, pst, P, ds->pixIndex, (const float4 *)c.taps, p0, nPixPass, ds->film);
    } else {
        const int fr = KZ_FILM_TILE + ftaps - 1;
        const size_t perSample = (size_t)(3 + 2 * ftaps) * fr * fr * sizeof(float);
        const int fchunk = (int)std::max<size_t>(1, std::min<size_t>({(size_t)Sp, (size_t)8, (size_t)(64 * 1024) / perSample}));
        const size_t fshm = perSample * fchunk;
        if (waitFilm) HIP_TRY(hipStreamWaitEvent(pst, waitFilm, 0));
        hipLaunchKernelGGL(kz_film_gather, fgrid, dim3(256), fshm, pst, P, ds->T.filter, ds->pixIndex, p0, nPixPass, Sp, fchunk, sJx, sJy, sR, sG, sB, ds->film);
    }
    HIP_TRY(hipGetLastError());
    return KZ_OK;
}

// Floats of the packed film rects of a tile list: tile t holds (h + 2b) x (w + 2b) x 4 floats - the tile with its filter apron.
size_t packedFloats(const KzParams &P, const KzTile *tiles, uint32_t nTiles) {
    size_t n = 0;
    for (uint32_t t = 0; t < nTiles; ++t) n += (size_t)(tiles[t].w + 2 * P.border) * (size_t)(tiles[t].h + 2 * P.border) * 4;
    return n;
}
int checkTiles(const KzParams &P, const KzTile *tiles, uint32_t nTiles) {
    if (!tiles && nTiles) return kz_fail(KZ_ERR_INVALID_ARG, "null tile list");
    for (uint32_t t = 0; t < nTiles; ++t) {
        const KzTile &tl = tiles[t];
        if (tl.x0 < 0 || tl.y0 < 0 || tl.w <= 0 || tl.h <= 0 || tl.x0 + tl.w > P.width || tl.y0 + tl.h > P.height)
            return kz_fail(KZ_ERR_INVALID_ARG, "tile %u (%d,%d %dx%d) outside the %dx%d image", t, tl.x0, tl.y0, tl.w, tl.h, P.width, P.height);
    }
    return KZ_OK;
}

// the film rects of `tiles` of replica ds -> host `packed` (through a device-side pack and a pinned staging buffer: one D2H copy at link rate)
int downloadTiles(KzScene *scene, KzDeviceState *ds, const KzTile *tiles, uint32_t nTiles, float *packed, size_t nFloats, hipStream_t stream) {
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
    if (nTiles > 65535u * 65535u) return kz_fail(KZ_ERR_UNSUPPORTED, "kz_film_download_tiles: %u tiles in one call", nTiles);
    const unsigned gz = (nTiles + 65534u) / 65535u, gy = gz > 1 ? 65535u : nTiles;
    hipLaunchKernelGGL(kz_film_pack, dim3((unsigned)maxRows, gy, gz), dim3(128), 0, stream, (const float4 *)ds->film, P.width + 2 * P.border, (const KzTileRect *)ds->rectsDev,
                       (const uint32_t *)ds->prevDev, P.border, ds->packDev, nTiles);
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

} // extern "C"
