// kz_plan.cpp - the pass planner (kz_plan.h): host arithmetic only. Every rule carries the measurement it comes from; the table in
// tests/test_plan_cpu.py pins what the rules give for the BASELINE configs, with and without a dealer, under tight limits, call after call.
#include "kz_plan.h"
#include "kz_internal.h"

#include <cstdio>

// The shape of a pass of `want` items over `nPixRange` pixels: sample count s, pixel count px.
static void shapeFor(const KzPlan &pl, size_t want, uint32_t nPixRange, uint32_t &s, uint32_t &px) {
    const uint32_t nSamples = pl.nSamples;
    want = std::max<size_t>(want, 64);
    // sppPerPass = n: n samples (or all the call asks for) of as many pixels as fit, pixel chunks in the order of the pixel list
    if (pl.sppPerPass > 0) { s = std::min<uint32_t>((uint32_t)pl.sppPerPass, nSamples); px = (uint32_t)std::min<size_t>(nPixRange, std::max<size_t>(64, want / s / 64 * 64)); return; }
    // default: every pixel of the range and as many samples as fit
    px = nPixRange; s = (uint32_t)std::min<size_t>(std::max<size_t>(1, want / std::max<uint32_t>(1, nPixRange)), nSamples);
    // A frame too large for 64 samples of every pixel per pass (C5 on one GPU: 16) is rendered in pixel chunks of 256 samples instead: a wave of the
    // camera-ray kernels is then one pixel again (one shared list), the film stage touches a chunk per pass instead of the whole frame, and the paths of a pass
    // stay in a part of the scene (C5, same call: 1 586 Msamples/s at 16 x all pixels, 1 708 at 64 x 2 M, 1 734 at 256 x 512 K).
    // (round 4, default pass size 2^30: C5 at 128 x all 8.3 M pixels 1 865 Msamples/s, at 256 x 4.2 M pixels 1 890: the chunks are taken below 256 samples then)
    const uint32_t chunkBelow = pl.autoShape ? 256u : 64u;
    if (s < chunkBelow && nSamples >= chunkBelow) { s = std::min<uint32_t>(256u, nSamples); px = (uint32_t)std::min<size_t>(nPixRange, std::max<size_t>(64, want / s / 64 * 64)); }
    // a multiple of 64 samples per pixel keeps every wave of the camera-ray kernels inside one pixel (one shared leaf list) - taken when it costs no extra pass
    // (a rank's share of a frame: 2^27 / 1 036 800 pixels = 129 -> 128)
    else if (s > 64 && s % 64 && (nSamples + s / 64 * 64 - 1) / (s / 64 * 64) == (nSamples + s - 1) / s) s = s / 64 * 64;
}

int kzPlanCall(const KzPlanIn &in, KzPlan &pl, std::string &err) {
    char buf[256];
    pl = KzPlan{};
    pl.nSamples = in.s1 - in.s0;
    pl.sppPerPass = in.sppPerPass;
    const uint32_t nSamples = pl.nSamples;
    const size_t perItem = std::max<size_t>(in.perItem, 1);
    // ---- pass geometry. A pass is pixPerPass pixels x S samples of each = up to passItems (pixel, sample) items (2^25 -> 981, 2^26 -> 1031, 2^27 -> 1061-1066,
    // 2^28 -> 1071 Msamples/s on C4 in round 1: fewer launches and shorter relative tails per sample).
    // Round 4, sized for the 288 GB of the card: with NOTHING said (passItems = passesInFlight = 0, no dealer) a call runs ONE pass at a time, as large as the state
    // budget allows up to 2^30 items (175 GB). Same-call sweeps (profiles/r04r_pass_size): C4 2 x 2^27 1 772, 2 x 2^28 1 804, 2 x 2^29 1 827, 1 x 2^30 1 849 Msamples/s; C5 1 821 -> 1 890;
    // C3 1 831 -> 1 842: fewer, longer kernels have shorter relative tails, and a second pass in flight buys less than the memory it takes is worth as pass size.
    // A dealer keeps two contexts (its batches overlap through them) of up to 2^29 items.
    pl.autoShape = in.pipeline == 2 && !in.passItems && !in.passesInFlight;
    int nCtx = in.pipeline == 2 ? (in.passesInFlight ? in.passesInFlight : (pl.autoShape && !in.dealer ? 1 : KZ_DEFAULT_PASSES_IN_FLIGHT)) : 1;
    // The default pass size (autoShape) is EARNED. A pass of 2^30 items is 5 % faster than passes of 2^27 (profiles/r04r_pass_size), but its 175 GB have a price
    // that someone pays: memory a process releases is wiped by the driver at ~33 GB/s, and whoever allocates before the wipe is through - the next job of a batch
    // of one-frame processes, a second process, this process's next scene - waits for the WHOLE wipe inside one allocation call, and the GPU work of that process
    // waits with it (profiles/r05a_alloc, r05d_cold_job). What a job CAN do is leave little behind. So the first call on a context asks for what rounds 1-3 ran
    // with - 2^28 items (47 GB: 1.4 s of wiping for whoever comes next, passes 3 % slower than 2^30) when the call has at least 2^30 items of work, 2^27 otherwise -
    // and a context may double with every further call: a one-frame job is over before the large pass would have paid, a process that keeps rendering
    // (bench.py's steps, scene after scene through the device's pool) runs passes of 2^30 from its third or fourth call.
    size_t earned = (size_t)1 << (in.dealer ? 29 : 30);
    if (pl.autoShape) {
        const size_t callItems = (size_t)in.nPix * nSamples / (in.dealer ? std::max<uint32_t>(1, in.takers) : 1u);
        const size_t byWork = (size_t)1 << (callItems >= ((size_t)1 << 30) ? 28 : 27);
        earned = std::min(earned, std::max(byWork, 2 * in.heldBefore));
    }
    pl.wantItems = std::max<size_t>(in.passItems ? (size_t)in.passItems : (pl.autoShape ? earned : (size_t)1 << 27), 64);
    pl.nPixSet = in.nPix;
    if (in.dealer) {
        if (!in.tilePixOffset || !in.nTiles) { err = "KzTileDealer: empty tile list"; return KZ_ERR_INVALID_ARG; }
        // a batch = about two passes' worth of (pixel, sample) items, never more, at most 1 / (4 x takers) of the list; counted with the LARGEST tile, and the pass
        // shape below with the largest batch. The pass size it is derived from must be the SAME on every taker of one counter: what a context has earned depends
        // on that device's history, so the batch is sized for what a dealer's context ends up with - 2^29 items - or for what the caller said.
        uint64_t maxTile = 1;
        for (uint32_t t = 0; t < in.nTiles; ++t) maxTile = std::max<uint64_t>(maxTile, in.tilePixOffset[t + 1] - in.tilePixOffset[t]);
        const uint64_t batchBasis = in.passItems ? in.passItems : (pl.autoShape ? (uint64_t)1 << 29 : (uint64_t)1 << 27);
        pl.batchTiles = in.dealerBatchTiles ? in.dealerBatchTiles
                      : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(2 * batchBasis / std::max<uint64_t>(1, maxTile * nSamples), std::max<uint32_t>(1, in.nTiles / (4 * std::max<uint32_t>(1, in.takers)))));
        pl.nPixSet = 1;                                          // (batches start at multiples of batchTiles: the counter only ever advances by that)
        for (uint32_t tb = 0; tb < in.nTiles; tb += pl.batchTiles) pl.nPixSet = std::max(pl.nPixSet, in.tilePixOffset[std::min(in.nTiles, tb + pl.batchTiles)] - in.tilePixOffset[tb]);
    }
    uint32_t S, pixPerPass;
    shapeFor(pl, pl.wantItems, pl.nPixSet, S, pixPerPass);
    // the largest pass of the wanted shape that fits `room` bytes: fewer samples first, then (from one sample) fewer pixels
    auto shape = [&](size_t room, uint32_t &s, uint32_t &px) {
        s = S; px = pixPerPass;
        if ((size_t)px * s * perItem <= room) return;
        const size_t sFit = room / px / perItem;
        if (sFit >= 1) { s = (uint32_t)std::min<size_t>(s, sFit); if (s > 64) s = s / 64 * 64; }      // (whole waves of one pixel for the camera-ray kernels)
        else { s = 1; px = (uint32_t)std::min<size_t>(px, room / perItem / 64 * 64); }
    };
    // as many contexts as wanted, but never more than there are passes (a call that is one pass runs it in one context at full size)
    uint32_t nPasses = 0;
    for (;; --nCtx) {
        uint32_t s, px;
        shape(in.limit / (size_t)nCtx, s, px);
        if (px > 0) nPasses = in.dealer ? 0xFFFFu : ((in.nPix + px - 1) / px) * ((nSamples + s - 1) / s);      // (a dealer: not known, assume many)
        if (nCtx > 1 && (px == 0 || nPasses < (uint32_t)nCtx)) continue;
        if (px == 0) {
            std::snprintf(buf, sizeof buf, "64 (pixel, sample) items need %zu bytes of path state, the limit is %zu", (size_t)64 * perItem, in.limit);
            err = buf; return KZ_ERR_OOM;
        }
        S = s; pixPerPass = px;
        break;
    }
    pl.S = S; pl.pixPerPass = pixPerPass; pl.nPasses = nPasses;
    pl.need = (size_t)pixPerPass * S;
    if (pl.need >= (1ull << 32)) { std::snprintf(buf, sizeof buf, "pass of %zu items (limit 2^32)", pl.need); err = buf; return KZ_ERR_UNSUPPORTED; }
    // Several passes in flight on internal streams when the call has at least two: the persistent traversal kernels of one pass
    // drain (fewer and fewer busy waves) while the other passes keep the machine full.
    pl.multi = in.pipeline == 2 && nPasses >= 2 && nCtx >= 2;
    pl.nCtx = pl.multi ? nCtx : 1;
    // A pass context GROWS (kz_arena.cpp): its memory is mapped on a side thread while the passes already run. With everything left to the library
    // (autoShape) a pass takes what its context holds at that moment - the first passes of a job that starts behind the driver's wipe of recently released
    // memory are small, on clean memory the context is complete before the first pass - and with an explicit pass size or number of passes in flight the
    // call waits for the size it was asked for. (Since round 6 the film does not depend on the pass structure either way.)
    // A call with everything left to the library also RUNS on what there is when its context cannot be completed - another process holds the card, or the arena
    // stops short of the reserve it leaves to the HIP runtime (kz_arena.cpp): narrower columns, more passes, the same film. It only starts EARLY (5 ms of grace per
    // level instead of waiting for the growth to end) when the context is large enough for the wait to matter.
    const bool onWhatThereIs = in.pipeline == 2 && pl.autoShape;
    pl.grow = onWhatThereIs && pl.need > ((size_t)1 << 26);
    pl.minStart = onWhatThereIs ? std::min<size_t>(pl.need, (size_t)1 << 20) : pl.need;
    pl.graceMs = pl.grow ? 5.0 : -1.0;
    return KZ_OK;
}

uint32_t kzPlanColumn(const KzPlan &pl, size_t avail, uint32_t nPixRange) {
    uint32_t w = std::min(pl.pixPerPass, nPixRange);
    if (avail < (size_t)w * std::min<uint32_t>(pl.S, pl.nSamples)) {             // the context is still growing (or holds less than it was asked for): the column it can serve now
        uint32_t sCol = 0, wCol = 0;
        shapeFor(pl, avail, nPixRange, sCol, wCol);
        w = std::max<uint32_t>(1, std::min(w, wCol));
    }
    // one sample of the column must fit what is mapped (shapeFor keeps every pixel of the range below 256 samples per pixel, whatever `avail` is)
    if ((size_t)w > avail) w = (uint32_t)std::max<size_t>(1, avail >= 64 ? avail / 64 * 64 : avail);
    return w;
}

uint32_t kzPlanSamples(const KzPlan &pl, size_t avail, uint32_t w, uint32_t remaining) {
    uint32_t Sp = (uint32_t)std::min<size_t>({(size_t)remaining, (size_t)pl.S, std::max<size_t>(1, avail / std::max<uint32_t>(1, w))});
    if (Sp > 64 && Sp < remaining) Sp = Sp / 64 * 64;                            // (whole waves of one pixel for the camera-ray kernels)
    return Sp;
}

// ---- the planner through the C ABI (kazen_mi355x_dev.h): what tests/test_plan_cpu.py tabulates ----
static void toIn(const KzPlanQuery *q, KzPlanIn &in) {
    in.pipeline = q->pipeline ? q->pipeline : 2; in.nPix = q->nPix; in.s0 = q->sampleBegin; in.s1 = q->sampleEnd;
    in.passItems = q->passItems; in.passesInFlight = q->passesInFlight; in.sppPerPass = q->sppPerPass;
    in.limit = (size_t)q->limitBytes; in.perItem = q->bytesPerItem ? (size_t)q->bytesPerItem : (size_t)176;
    in.dealer = q->dealer != 0; in.takers = q->takers; in.dealerBatchTiles = q->batchTiles; in.nTiles = q->nTiles; in.tilePixOffset = q->tilePixOffset;
    in.heldBefore = (size_t)q->heldItems;
}

extern "C" {

int kz_plan_passes(const KzPlanQuery *q, KzPlanAnswer *a) {
    if (!q || !a) return kz_fail(KZ_ERR_INVALID_ARG, "kz_plan_passes: null argument");
    if (q->sampleBegin >= q->sampleEnd || !q->nPix) return kz_fail(KZ_ERR_INVALID_ARG, "kz_plan_passes: empty call");
    KzPlanIn in; toIn(q, in);
    KzPlan pl; std::string err;
    if (const int rc = kzPlanCall(in, pl, err)) return kz_fail(rc, "%s", err.c_str());
    *a = KzPlanAnswer{};
    a->autoShape = pl.autoShape; a->nCtx = pl.nCtx; a->multi = pl.multi; a->grow = pl.grow; a->S = pl.S; a->pixPerPass = pl.pixPerPass;
    a->batchTiles = pl.batchTiles; a->nPixSet = pl.nPixSet; a->nPasses = pl.nPasses; a->need = pl.need; a->wantItems = pl.wantItems;
    a->minStart = pl.minStart; a->graceMs = pl.graceMs;
    return KZ_OK;
}

int kz_plan_schedule(const KzPlanQuery *q, const uint64_t *avail, uint32_t nAvail, uint32_t pixBegin, uint32_t pixEnd, uint32_t *passes, uint32_t cap, uint32_t *nPasses) {
    if (!q || !avail || !nAvail || !nPasses || (cap && !passes)) return kz_fail(KZ_ERR_INVALID_ARG, "kz_plan_schedule: null argument");
    if (q->sampleBegin >= q->sampleEnd || pixBegin >= pixEnd) return kz_fail(KZ_ERR_INVALID_ARG, "kz_plan_schedule: empty call");
    KzPlanIn in; toIn(q, in);
    KzPlan pl; std::string err;
    if (const int rc = kzPlanCall(in, pl, err)) return kz_fail(rc, "%s", err.c_str());
    uint32_t asked = 0, n = 0;
    size_t last = 0;
    const int rc = kzPlanRun(pl, pixBegin, pixEnd, q->sampleBegin, q->sampleEnd,
        [&](size_t *usable) { last = (size_t)avail[std::min(asked, nAvail - 1)]; ++asked; *usable = std::min(last, pl.need); return last ? 0 : KZ_ERR_OOM; },
        [&](uint32_t p0, uint32_t w, uint32_t s, uint32_t Sp) {
            if ((size_t)w * Sp > last) return (int)KZ_ERR_STATE;                    // a pass beyond what its context holds: the planner's one invariant
            if (n < cap) { passes[4 * n] = p0; passes[4 * n + 1] = w; passes[4 * n + 2] = s; passes[4 * n + 3] = Sp; }
            ++n;
            return 0;
        });
    *nPasses = n;
    if (rc == KZ_ERR_STATE) return kz_fail(rc, "kz_plan_schedule: a pass of more items than its context holds");
    if (rc) return kz_fail(rc, "kz_plan_schedule: a context holds nothing");
    return KZ_OK;
}

} // extern "C"
