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
// a25 ImageBlock::put (block.cpp:56-96) as RUNNING TAP SUMS (round 6). For every pixel of the frame the replica keeps taps x taps accumulators
// (rgb * w, w) - one per film texel the pixel's samples can reach - laid out tap-major over the frame, [tap][y * width + x]:
//   kz_film_taps      after every pass: one wave per 64 / GROUPS consecutive pixels of the pass's pixel list. It LOADS the pixel's accumulators, adds the
//                     pass's samples of that pixel in sample order - validity (Color3f::isValid), the separable filter weights of block.cpp:64-80 per tap,
//                     the weighted products of block.cpp:84 - and stores them back. An accumulator is therefore the sum, in sample order, of everything
//                     the pixel has been given since the film was cleared, WHATEVER passes, calls, contexts or batches brought the samples: the grouping of
//                     the float additions no longer depends on the pass size, on how fast the pass context grew, on the number of passes in flight (H10).
//   kz_film_resolve   once per call: film texel = sum, over the cells of the canonical tile grid (G x G pixels, row-major) its footprint touches, of the
//                     cell's partial sum (taps in (row, column) order). This is exactly what the host merge of per-tile rects forms (ImageBlock::put(ImageBlock&)
//                     in tile order), so the film of one device, of N devices with tiles dealt beforehand and of N devices taking batches from a counter are
//                     the same bits when the tiles are the grid's.
//   kz_film_tile_rects the packed rect of a tile (the tile with its filter apron, the extent of an ImageBlock): the partial sums of THAT tile's pixels only.
// The weights are formed block-relative exactly as the reference does (32x32 blocks at multiples of KAZEN_BLOCK_SIZE), so they are bit-identical to
// ImageBlock::put; only the order of the float additions is the build's own (fixed) one.
// ============================================================================================
#define KZ_TAPS_CHUNK1 8                     // samples per staging round with one lane per pixel (32-B pieces of every sample row; round 2's kernel)
#define KZ_TAPS_CHUNK 16                     // ... with several lanes per pixel: 64-B pieces = whole HBM sectors (profiles/r05h_film_taps)
// GROUPS lane groups share a pixel: group g carries the tap ROWS [g * R0, g * R0 + R0) of its 64 / GROUPS pixels, R0 = ceil(TAPS / GROUPS). Every accumulator
// receives the same products in the same (sample) order whatever GROUPS is; what is paid for more groups is the per-sample set-up (validity, footprint, the
// x weights) once per group. TAPS <= 5 (every default of the reference): two groups - 15 and 10 accumulators instead of 25, 10.5 KB of staging per wave
// (round 5: film stage of a 2^30-item pass of C4 12.6 -> 10.5 ms); 6 .. 9 taps (radius up to 4): four groups, at most 27 accumulators per lane.
template <int TAPS, int GROUPS>
__global__ __launch_bounds__(64) void kz_film_taps(KzParams P, const float *__restrict__ filter, const uint32_t *__restrict__ pixList, uint32_t nPix, uint32_t S,
                                                   const float *__restrict__ inJx, const float *__restrict__ inJy, const float *__restrict__ inR,
                                                   const float *__restrict__ inG, const float *__restrict__ inB, float4 *__restrict__ tapSums, size_t framePix) {
    constexpr int PIX = 64 / GROUPS, R0 = (TAPS + GROUPS - 1) / GROUPS, CHUNK = GROUPS == 1 ? KZ_TAPS_CHUNK1 : KZ_TAPS_CHUNK;
    __shared__ float s_filter[KZ_FILTER_RESOLUTION + 1];
    __shared__ float s_in[5][CHUNK][PIX + 1];                          // [array][sample][pixel], rows padded against bank conflicts of the transposing store
    const int lane = threadIdx.x, pixLane = lane % PIX, group = lane / PIX;
    if (lane <= KZ_FILTER_RESOLUTION) s_filter[lane] = filter[lane];
    const uint32_t pl0 = blockIdx.x * (uint32_t)PIX, pl = pl0 + (uint32_t)pixLane;
    const bool havePixel = pl < nPix;
    const uint32_t pxy = havePixel ? pixList[pl] : 0u;
    const int px = (int)(pxy & 0xffffu), py = (int)(pxy >> 16);
    const int bx0 = px & ~31, by0 = py & ~31;                          // the reference block this pixel is rendered in
    const float r = P.filterRadius, lf = P.lookupFactor;
    const int ty0 = group * R0, nRowsMine = min(R0, max(0, TAPS - ty0));
    float xb[TAPS], yb[R0];                                            // block-relative film coordinates this pixel reaches through tap t
#pragma unroll
    for (int t = 0; t < TAPS; ++t) xb[t] = (float)(px + P.border - P.tapLo - t - bx0);
#pragma unroll
    for (int t = 0; t < R0; ++t) yb[t] = (float)(py + P.border - P.tapLo - (ty0 + t) - by0);
    float4 *const mine = tapSums + (size_t)py * (size_t)P.width + (size_t)px;
    float4 acc[R0 * TAPS];
#pragma unroll
    for (int ty = 0; ty < R0; ++ty)
#pragma unroll
        for (int tx = 0; tx < TAPS; ++tx)
            acc[ty * TAPS + tx] = (havePixel && ty < nRowsMine) ? mine[(size_t)((ty0 + ty) * TAPS + tx) * framePix] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float *const in[5] = {inJx, inJy, inR, inG, inB};
    const uint32_t nRows = min((uint32_t)PIX, nPix - min(nPix, pl0));  // pixels of this wave
    for (uint32_t c0 = 0; c0 < S; c0 += CHUNK) {
        const uint32_t n = min((uint32_t)CHUNK, S - c0);
        __syncthreads();
        for (uint32_t i = lane; i < nRows * CHUNK; i += 64u) {         // consecutive lanes: consecutive samples of one pixel, then the next pixel
            const uint32_t p = i / CHUNK, k = i % CHUNK;
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
                const float cr = s_in[2][k][pixLane], cg = s_in[3][k][pixLane], cb = s_in[4][k][pixLane];
                const bool valid = cr >= 0.f && cg >= 0.f && cb >= 0.f && isfinite(cr) && isfinite(cg) && isfinite(cb);   // Color3f::isValid
                if (!valid) continue;                                  // an invalid sample carries weight 0 everywhere: adds exact zeros
                const float posx = ((float)px + jx) - 0.5f - (float)(bx0 - P.border), posy = ((float)py + jy) - 0.5f - (float)(by0 - P.border);   // block.cpp:64-67
                const float lox = ceilf(posx - r), hix = floorf(posx + r), loy = ceilf(posy - r), hiy = floorf(posy + r);                     // block.cpp:70-73
                float wx[TAPS], wy[R0];
#pragma unroll
                for (int t = 0; t < TAPS; ++t) wx[t] = !(xb[t] < lox || xb[t] > hix) ? s_filter[(int)(fabsf(xb[t] - posx) * lf)] : 0.f;          // block.cpp:77-80
#pragma unroll
                for (int t = 0; t < R0; ++t) wy[t] = (t < nRowsMine && !(yb[t] < loy || yb[t] > hiy)) ? s_filter[(int)(fabsf(yb[t] - posy) * lf)] : 0.f;
#pragma unroll
                for (int ty = 0; ty < R0; ++ty)
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
        for (int ty = 0; ty < R0; ++ty)
            if (ty < nRowsMine) {
#pragma unroll
                for (int tx = 0; tx < TAPS; ++tx) mine[(size_t)((ty0 + ty) * TAPS + tx) * framePix] = acc[ty * TAPS + tx];
            }
    }
}

// The partial sum a film texel receives from the source pixels [xa, xb] x [ya, yb] (all inside the texel's footprint, whose first source column / row is
// x0s / y0s): taps in (row, column) order. Shared by the two kernels below - they must form the same bits.
__device__ __forceinline__ float4 kzTapPartial(const float4 *__restrict__ tapSums, size_t framePix, int width, int taps, int x0s, int y0s, int xa, int xb, int ya, int yb) {
    float4 part = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y = ya; y <= yb; ++y)
        for (int x = xa; x <= xb; ++x) {
            const float4 t = tapSums[(size_t)((y - y0s) * taps + (x - x0s)) * framePix + (size_t)y * (size_t)width + (size_t)x];
            part.x += t.x; part.y += t.y; part.z += t.z; part.w += t.w;
        }
    return part;
}

// film texel = sum over the grid cells its footprint touches (row-major) of that cell's partial: ImageBlock::put(ImageBlock&) (block.cpp:87-96) of the
// cells' blocks in tile order. Every texel of the film is written (zero where nothing reaches it).
__global__ __launch_bounds__(256) void kz_film_resolve(KzParams P, const float4 *__restrict__ tapSums, size_t framePix, int G, float4 *__restrict__ film) {
    const int cols = P.width + 2 * P.border, rows = P.height + 2 * P.border;
    const int fx = blockIdx.x * 16 + (threadIdx.x & 15), fy = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (fx >= cols || fy >= rows) return;
    const int taps = P.tapHi - P.tapLo + 1;
    const int x0s = fx - P.border + P.tapLo, y0s = fy - P.border + P.tapLo;      // the source pixel that reaches this texel through tap (0, 0)
    const int xlo = max(x0s, 0), xhi = min(x0s + taps - 1, P.width - 1), ylo = max(y0s, 0), yhi = min(y0s + taps - 1, P.height - 1);
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xlo <= xhi && ylo <= yhi) {
        for (int tr = ylo / G; tr <= yhi / G; ++tr)
            for (int tc = xlo / G; tc <= xhi / G; ++tc) {
                const float4 part = kzTapPartial(tapSums, framePix, P.width, taps, x0s, y0s, max(xlo, tc * G), min(xhi, tc * G + G - 1), max(ylo, tr * G), min(yhi, tr * G + G - 1));
                total.x += part.x; total.y += part.y; total.z += part.z; total.w += part.w;
            }
    }
    film[(size_t)fy * cols + fx] = total;
}

// The film rects of a tile list, packed: tile t contributes its (h + 2b) x (w + 2b) rect (the tile with its filter apron) as consecutive rows at
// rects[t].offset (in float4s), each texel = what the TILE'S OWN pixels add to it - an ImageBlock of the tile. The aprons of neighbouring tiles overlap in
// the film; the merge adds them there. One workgroup per (tile, row).
__global__ __launch_bounds__(128) void kz_film_tile_rects(KzParams P, const float4 *__restrict__ tapSums, size_t framePix, const KzTileRect *__restrict__ rects,
                                                          float4 *__restrict__ out, uint32_t nTiles) {
    const uint32_t tile = blockIdx.y + blockIdx.z * 65535u;             // (a grid's y extent ends at 65535: longer tile lists continue in z)
    if (tile >= nTiles) return;
    const KzTileRect r = rects[tile];
    const int rw = r.w + 2 * P.border, rh = r.h + 2 * P.border;
    const int row = blockIdx.x;
    if (row >= rh) return;
    const int taps = P.tapHi - P.tapLo + 1;
    const int fy = r.y0 + row, y0s = fy - P.border + P.tapLo;
    const int ya = max(y0s, r.y0), yb = min(y0s + taps - 1, r.y0 + r.h - 1);
    float4 *dst = out + r.offset + (size_t)row * rw;
    for (int x = threadIdx.x; x < rw; x += blockDim.x) {
        const int x0s = r.x0 + x - P.border + P.tapLo;
        dst[x] = kzTapPartial(tapSums, framePix, P.width, taps, x0s, y0s, max(x0s, r.x0), min(x0s + taps - 1, r.x0 + r.w - 1), ya, yb);
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

// The replica's running tap sums: taps^2 float4 per pixel of the frame (C4, 5 taps: 829 MB; C5: 3.3 GB), allocated and zeroed on first use.
int kzFilmEnsureTapSums(KzScene *scene, KzDeviceState *ds, hipStream_t stream) {
    const KzParams &P = scene->prm;
    const size_t taps = (size_t)(P.tapHi - P.tapLo + 1), framePix = (size_t)P.width * (size_t)P.height;
    if (ds->tapSums) return KZ_OK;
    KZ_TRACE("film: tap sums, %.0f MB ...", taps * taps * framePix * sizeof(float4) / 1e6);
    KZ_ALLOC(&ds->tapSums, taps * taps * framePix * sizeof(float4));
    ds->tapSumsBytes = taps * taps * framePix * sizeof(float4);
    HIP_TRY(hipMemsetAsync(ds->tapSums, 0, ds->tapSumsBytes, stream));
    KZ_TRACE("film: ... there");
    return KZ_OK;
}

int kzFilmClear(KzDeviceState *ds, hipStream_t stream) {
    if (ds->tapSums) HIP_TRY(hipMemsetAsync(ds->tapSums, 0, ds->tapSumsBytes, stream));
    HIP_TRY(hipMemsetAsync(ds->film, 0, ds->filmPixels * sizeof(float4), stream));
    return KZ_OK;
}

// The film stage of one pass (called by renderOn, kz_render.hip): ImageBlock::put for every sample record of the pass, into the running tap sums of the pass's
// pixels. The kernel reads and writes those sums, so it waits for the film stage of the pass before it (`waitFilm`, passes on other streams; null: same stream).
// lanesPerPixel: 0 = default (2 lane groups per pixel up to 5 taps, 4 beyond), 1 = one lane per pixel (round 2's kernel; up to 5 taps) - the same sums bit for bit.
int kzFilmStage(KzScene *scene, KzDeviceState *ds, PassCtx &c, hipStream_t pst, const uint32_t *pixList, uint32_t nPixPass, uint32_t Sp, hipEvent_t waitFilm, int lanesPerPixel) {
    const KzParams &P = scene->prm;
    const int ftaps = P.tapHi - P.tapLo + 1;
    const size_t framePix = (size_t)P.width * (size_t)P.height;
    float *sJx = c.plane[0], *sJy = c.plane[1], *sR = c.plane[2], *sG = c.plane[3], *sB = c.plane[4];
    if (waitFilm) HIP_TRY(hipStreamWaitEvent(pst, waitFilm, 0));
#define KZ_FILM_TAPS(N, GR) hipLaunchKernelGGL((kz_film_taps<N, GR>), dim3((nPixPass + 64 / GR - 1) / (64 / GR)), dim3(64), 0, pst, P, ds->T.filter, pixList, nPixPass, Sp, sJx, sJy, sR, sG, sB, ds->tapSums, framePix)
    if (lanesPerPixel == 1 && ftaps <= 5) switch (ftaps) { case 1: KZ_FILM_TAPS(1, 1); break; case 2: KZ_FILM_TAPS(2, 1); break; case 3: KZ_FILM_TAPS(3, 1); break; case 4: KZ_FILM_TAPS(4, 1); break; default: KZ_FILM_TAPS(5, 1); break; }
    else switch (ftaps) {
        case 1: KZ_FILM_TAPS(1, 2); break; case 2: KZ_FILM_TAPS(2, 2); break; case 3: KZ_FILM_TAPS(3, 2); break; case 4: KZ_FILM_TAPS(4, 2); break; case 5: KZ_FILM_TAPS(5, 2); break;
        case 6: KZ_FILM_TAPS(6, 4); break; case 7: KZ_FILM_TAPS(7, 4); break; case 8: KZ_FILM_TAPS(8, 4); break; case 9: KZ_FILM_TAPS(9, 4); break;
        default: return kz_fail(KZ_ERR_UNSUPPORTED, "film filter of %d taps per axis (at most %d)", ftaps, KZ_MAX_FILTER_TAPS);
    }
#undef KZ_FILM_TAPS
    HIP_TRY(hipGetLastError());
    return KZ_OK;
}

// The film of the replica from its tap sums (once per call, on the call's stream behind every pass): ImageBlock::put(ImageBlock&) of the canonical grid's tiles in tile order.
int kzFilmResolve(KzScene *scene, KzDeviceState *ds, hipStream_t stream) {
    const KzParams &P = scene->prm;
    const int cols = P.width + 2 * P.border, rows = P.height + 2 * P.border;
    hipLaunchKernelGGL(kz_film_resolve, dim3((cols + 15) / 16, (rows + 15) / 16), dim3(256), 0, stream, P, (const float4 *)ds->tapSums, (size_t)P.width * (size_t)P.height, KZ_FILM_GRID, ds->film);
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
    size_t off = 0; int maxRows = 0;
    for (uint32_t t = 0; t < nTiles; ++t) {
        rects[t] = KzTileRect{tiles[t].x0, tiles[t].y0, tiles[t].w, tiles[t].h, (uint32_t)off};
        off += (size_t)(tiles[t].w + 2 * P.border) * (size_t)(tiles[t].h + 2 * P.border);
        maxRows = std::max(maxRows, tiles[t].h + 2 * P.border);
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
    HIP_TRY(hipMemcpyAsync(ds->rectsDev, rects.data(), (size_t)nTiles * sizeof(KzTileRect), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));                              // (the tables are host vectors of this call)
    if (nTiles > 65535u * 65535u) return kz_fail(KZ_ERR_UNSUPPORTED, "kz_film_download_tiles: %u tiles in one call", nTiles);
    const unsigned gz = (nTiles + 65534u) / 65535u, gy = gz > 1 ? 65535u : nTiles;
    if ((rc = kzFilmEnsureTapSums(scene, ds, stream))) return rc;      // (a replica nothing has been rendered on: zeros)
    hipLaunchKernelGGL(kz_film_tile_rects, dim3((unsigned)maxRows, gy, gz), dim3(128), 0, stream, P, (const float4 *)ds->tapSums, (size_t)P.width * (size_t)P.height, (const KzTileRect *)ds->rectsDev,
                       ds->packDev, nTiles);
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
    return kzFilmClear(ds, (hipStream_t)stream);
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
