// kz_plan.h - the pass planner of kz_render / kz_render_tiles: pure host arithmetic, no HIP call, no state (kz_plan.cpp). A call renders
// nPix pixels x the sample indices [s0, s1) in PASSES of pixPerPass pixels x S samples = up to `need` (pixel, sample) items each, in nCtx pass
// contexts; this unit decides those numbers from what the caller asked for, the memory it may use and what the first context held before
// (the default pass size is EARNED call by call, DESIGN.md "pass policy"), and walks the schedule "pixel columns x sample ranges" against
// what a context that may still be growing holds at the moment a pass is planned. renderOn (kz_render.hip) is plan -> ensure -> launch.
// tests/test_plan_cpu.py drives the same code through kz_plan_passes / kz_plan_schedule (kazen_mi355x_dev.h) on the CPU box: its table is
// the documentation of the policy.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/kazen_mi355x.h"

struct KzPlanIn {
    int pipeline = 2;                 // 1 = megakernel, 2 = wavefront
    uint32_t nPix = 0;                // pixels of the call's tile set
    uint32_t s0 = 0, s1 = 0;          // sample indices [s0, s1)
    uint64_t passItems = 0;           // KzRenderOpts.passItems (0 = default)
    int passesInFlight = 0;           // KzRenderOpts.passesInFlight (0 = default)
    int sppPerPass = 0;               // KzTuning.sppPerPass (0 = default)
    size_t limit = 0;                 // bytes the pass contexts of this call may hold together
    size_t perItem = 1;               // bytes of path state + sample record per item
    // dynamic dealing (KzTileDealer): the tile set is the whole list, a batch of tiles is a range of its pixel list
    bool dealer = false;
    uint32_t takers = 1, dealerBatchTiles = 0, nTiles = 0;
    const uint32_t *tilePixOffset = nullptr;      // nTiles + 1 entries: first list position of every tile, then the total
    // history: items the first context was last asked to hold / holds (a fresh replica: the largest context in the device's pool)
    size_t heldBefore = 0;
};

struct KzPlan {
    bool autoShape = false;           // everything left to the library: ONE pass at a time, its size earned call by call, contexts that grow while passes run
    int nCtx = 1;                     // pass contexts (= passes in flight) this call uses
    bool multi = false;               // passes on internal streams (nCtx >= 2 and at least two passes)
    uint32_t S = 1, pixPerPass = 0;   // the target shape of a pass
    size_t need = 0;                  // = pixPerPass x S: items a context is asked to hold
    size_t wantItems = 0;             // the pass size aimed at before the memory limit was applied
    uint32_t nPasses = 0;             // passes of the call if every context is complete (a dealer: unknown, 0xFFFF)
    uint32_t nSamples = 0, nPixSet = 0;   // nPixSet: pixels a pass shape is chosen for (the tile set; with a dealer the largest batch)
    uint32_t batchTiles = 0;          // dealer: tiles per batch
    bool grow = false;                // passes may start on a context that is still growing (autoShape, large passes)
    size_t minStart = 0;              // items a context must hold before the first pass is planned on it
    double graceMs = -1.0;            // how long a slow growth step is waited for before a pass goes ahead with what there is (< 0: wait for `need`)
    int sppPerPass = 0;
};

// Fails (KZ_ERR_* code, message in err) when not even 64 items fit the limit or a pass would exceed 2^32 items.
int kzPlanCall(const KzPlanIn &in, KzPlan &pl, std::string &err);
// The pixel column a context holding `avail` items can serve now out of `nPixRange` remaining pixels: never wider than the target shape, never more
// pixels than `avail` items (a pass of one sample of the column must fit what is MAPPED: the ranges beyond are reserved address space, not memory).
uint32_t kzPlanColumn(const KzPlan &pl, size_t avail, uint32_t nPixRange);
// Samples of the next pass of a column of `w` pixels with `remaining` samples to go, in a context holding `avail` >= w items.
uint32_t kzPlanSamples(const KzPlan &pl, size_t avail, uint32_t w, uint32_t remaining);

// The passes of pixels [b0, b1) of the pixel list x sample indices [sFrom, s1): columns of pixels, each rendered in ascending sample ranges (the film's
// running tap sums need nothing else: any pass structure gives the same film). nextCtx(&avail) prepares the context of the NEXT pass and tells what it
// holds now; onePass(p0, w, s, Sp) launches it. When the context of a later sample range holds less than one sample of the column (another context than
// the one the column was sized for; growth that failed), the rest of the column is finished in narrower columns.
template <class NextCtx, class OnePass>
int kzPlanRun(const KzPlan &pl, uint32_t b0, uint32_t b1, uint32_t sFrom, uint32_t s1, NextCtx &&nextCtx, OnePass &&onePass) {
    int rc;
    for (uint32_t p0 = b0; p0 < b1;) {
        size_t avail = 0;
        if ((rc = nextCtx(&avail))) return rc;
        const uint32_t w = kzPlanColumn(pl, avail, b1 - p0);
        for (uint32_t s = sFrom; s < s1;) {
            if (s != sFrom && (rc = nextCtx(&avail))) return rc;
            if (avail < w) { if ((rc = kzPlanRun(pl, p0, p0 + w, s, s1, nextCtx, onePass))) return rc; break; }
            const uint32_t Sp = kzPlanSamples(pl, avail, w, s1 - s);
            if ((rc = onePass(p0, w, s, Sp))) return rc;
            s += Sp;
        }
        p0 += w;
    }
    return 0;
}

// ---- dynamic dealing: the two operations on the words the takers of one KzTileDealer share (they may live in memory shared between processes). Host code
// only, so that the multi-device driver's threads can be run against them under ThreadSanitizer without a GPU (tests/host_cpp/multi_tsan_test.cpp).
// The takers of one counter must have resolved the same batch size for the same list: the first publishes what it resolved in `agreed`, a taker that
// resolved anything else learns it before it takes a tile. False = disagreement.
inline bool kzDealerAgree(const KzTileDealer *d, uint32_t batchTiles, uint32_t nTiles) {
    if (!d->agreed) return true;
    const uint32_t mine = ((batchTiles * 0x9E3779B1u) ^ (nTiles * 0x85EBCA6Bu)) | 1u;
    uint32_t seen = 0;
    return __atomic_compare_exchange_n((uint32_t *)d->agreed, &seen, mine, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE) || seen == mine;
}
// BlockGenerator::next (block.cpp:117-148): the next batch [tb, te) of the tile list; false when the list is dealt or the caller's `taken` buffer is full.
inline bool kzDealerTake(const KzTileDealer *d, uint32_t batchTiles, uint32_t nTiles, uint32_t &tb, uint32_t &te) {
    if (d->takenCap && *d->nTaken + 2 > d->takenCap) return false;
    tb = __atomic_fetch_add((uint32_t *)d->counter, batchTiles, __ATOMIC_RELAXED);
    if (tb >= nTiles) return false;
    te = std::min(nTiles, tb + batchTiles);
    if (d->takenCap) { d->taken[*d->nTaken] = tb; d->taken[*d->nTaken + 1] = te; *d->nTaken += 2; }
    return true;
}
