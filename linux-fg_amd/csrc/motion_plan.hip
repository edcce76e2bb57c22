// motion_plan.hip -- host side of the prefiltered motion path: how a frame's 56 x 64 tiles become work units for the persistent
// kernel (motion_prefilter.hip) and the lean kernel (motion_lean.hip), and where everything lives in a lane's workspace.
// Replaces nothing of the reference's (src/frame_manager.cpp:342-343 dispatches one workgroup per 16 x 16 pixels and is done).
#include "lfg_motion_common.hpp"

#include <algorithm>
#include <utility>
#include <vector>

namespace lfg {

size_t motion_workspace_bytes(uint32_t width, uint32_t height, int slots, int rimSplit, int rimSplit2, MotionWorkspaceLayout *layout) {
    const size_t px = (size_t)width * height;
    const size_t tiles = (size_t)((width + kTW - 1) / kTW) * ((height + kTH - 1) / kTH);
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    MotionWorkspaceLayout l;
    l.list = 0;
    l.umin = align(l.list + px * kListK * sizeof(Rec));
    l.count = align(l.umin + px * sizeof(float));
    l.tileFlags = align(l.count + px * sizeof(uint32_t));
    // one word per 16-row segment of a prefilter tile, right behind the flags: one memset clears both
    const size_t ptiles = (size_t)((width + kPTW - 1) / kPTW) * ((height + kPTH - 1) / kPTH);
    // ... and behind them the map of the segments handed over at run time and the length of their queue
    l.segDone = l.tileFlags + tiles * sizeof(uint32_t);
    l.segMap = l.segDone + ptiles * (kPTH / kSeg) * sizeof(uint32_t);
    l.queueCount = l.segMap + ptiles * (kPTH / kSeg) * sizeof(uint32_t);
    // (+ the number of flagged tiles, the list of the first kShareBelow of them and a counter of arrived parts for each,
    //  + the prefilter's three unit counters and the hint kernel's count of finished workgroups)
    l.ctrl = l.queueCount + (2 + 2 * kShareBelow) * sizeof(uint32_t);
    // ... and the queue of segments handed over at run time (entries double as "slot filled" signals): up to a quarter
    // of the frame's segments, 2048 entries at most (a multiple of the entries one segment takes, so that a push either fits as
    // a whole or is refused as a whole; beyond that a segment is searched by the wave that owns it, as before).  One
    // memset clears everything from the tile flags to here.
    l.queueCap = (int)std::min<size_t>(2048, std::max<size_t>(LFG_DYN_PARTS / 4, ptiles * (kPTH / kSeg) / 4 / (LFG_DYN_PARTS / 4) * (LFG_DYN_PARTS / 4)));
    l.queue = l.ctrl + 8 * sizeof(uint32_t);     // ctrl: [0..2] the prefilter's unit counters, [4] open segments
    l.colBand = l.queue + (size_t)l.queueCap * sizeof(uint32_t);                 // the strip kernel's tables (motion_strip.hip): cleared with the rest
    l.rowBand = l.colBand + (size_t)height * sizeof(uint32_t);
    l.order = align(l.rowBand + (size_t)width * sizeof(uint32_t));              // this call's hints and visiting order              // this call's hints and visiting order
    // work-unit tables and the auxiliary arrays of the shared tiles (see prefilter_plan): one 56 x 64 block per unit
    // (two plans side by side where the lean kernel may run -- frames in flight: the one for calls that go through it, rimSplit2,
    //  and the one for calls that do not; the host picks per call, lfg_capi.cpp: motion_run.  The auxiliary arrays serve either.)
    const PrefilterPlanHost plan = prefilter_plan(width, height, slots, rimSplit);
    const PrefilterPlanHost plan2 = rimSplit2 ? prefilter_plan(width, height, slots, rimSplit2) : PrefilterPlanHost();
    const size_t auxUnits = (size_t)std::max(plan.auxUnits, plan2.auxUnits);
    l.orderFlags = l.order + (size_t)(kHints + kCand) * sizeof(uint32_t);          // hand-over allowed | hints in front | the verdict word
    l.verdict = l.orderFlags + 2 * sizeof(uint32_t);
    l.plan = align(l.order + (kHints + kCand + 3) * sizeof(uint32_t));
    l.plan2 = align(l.plan + (size_t)(2 * plan.units + plan.tiles) * sizeof(uint32_t));
    l.units = plan.units; l.units2 = plan2.units; l.units2Static = plan2.units - (int)plan2.leanTiles.size(); l.rimSplit2 = rimSplit2; l.tiles = plan.tiles;
    l.auxList = align(l.plan2 + (size_t)(2 * plan2.units + plan2.tiles) * sizeof(uint32_t));
    l.auxUmin = align(l.auxList + auxUnits * kPTH * kPTW * kListAux * sizeof(Rec));
    l.auxCount = align(l.auxUmin + auxUnits * kPTH * kPTW * sizeof(float));
    // each queue entry owns four 16-row blocks of private lists (0.92 MB)
    const size_t dynBlocks = (size_t)l.queueCap * 4;
    l.dynList = align(l.auxCount + auxUnits * kPTH * kPTW * sizeof(uint32_t));
    l.slots = slots;
    l.rimSplit = rimSplit;
    l.listMain = kListK; l.listAux = kListAux; l.listDyn = kListDyn;
    l.dynUmin = align(l.dynList + dynBlocks * kSeg * kPTW * kListDyn * sizeof(Rec));
    l.dynCount = align(l.dynUmin + dynBlocks * kSeg * kPTW * sizeof(float));
    l.dynInit = align(l.dynCount + dynBlocks * kSeg * kPTW * sizeof(uint32_t));
    // where the parts of a flagged tile meet (motion_tiled_8_16_kernel): kShareBelow slots of 64 x 64 words, all ones between calls
    // the segments the prefilter leaves open, for the resolve kernel: one word per 16-row segment at most
    l.openList = align(l.dynInit + (size_t)l.queueCap / (LFG_DYN_PARTS / 4) * kSeg * kPTW * sizeof(uint32_t));      // (one block per segment: see the push)
    l.merge = align(l.openList + ptiles * (kPTH / kSeg) * sizeof(uint32_t));
    l.mergeBytes = (size_t)kShareBelow * kTW * kTH * sizeof(unsigned long long);
    l.leanTiles = align(l.merge + l.mergeBytes);
    l.leanCount = (int)plan2.leanTiles.size();
    l.leanLaunch = plan2.leanTiles.empty() ? 0 : (int)(plan2.leanTiles.size() + plan2.leanPartial.size());
    l.hardTiles = align(l.leanTiles + (plan2.leanTiles.size() + plan2.leanPartial.size() + 1) * sizeof(uint32_t));
    l.total = align(l.hardTiles + (plan2.leanTiles.size() + 1) * sizeof(uint32_t));
    if (layout) *layout = l;
    return l.total;
}

// How the prefilter's tiles become work units for `slots` concurrently resident workgroups (unitMap entry:
// tile | first chunk << 20 | chunks << 24 | segment unit << 28 | segment << 29).
//   * Tiles on the rim of the image (some block position outside it) hold pixels without a good match -- the band a
//     moving camera exposes, the rows and columns an upscaler filters differently at the edge -- and the segments
//     that contain them run the full search, hundreds of times the work of a segment in which the partial-distortion
//     test fires.  A rim tile therefore becomes one unit per SEGMENT, whose four waves take a quarter of the
//     candidate order each (private lists, merged by the resolve kernel): the waves of a workgroup finish together
//     whether their segment is a cheap or an expensive one.  These units are dispatched first, so that the long
//     ones start at once and the short ones fill in behind them.
//   * A frame with fewer tiles than half the slots has every tile shared by up to 8 units (contiguous parts of the
//     candidate order, private lists), to fill the chip.
//   * The parts of a rim segment, 4 or 8 (rimSplit: lfg_capi.cpp, motion_rim_split -- four unless LFG_MOTION_RIM_SPLIT=8 asks;
//     the measurements are there).
PrefilterPlanHost prefilter_plan(uint32_t width, uint32_t height, int slots, int rimSplit) {
    // 4 or 8 parts of the order per rim segment; 1: rim tiles whole; 48: four, and eight for the segments whose position rows
    // leave the image at its top or bottom -- the strip a vertical pan exposes is searched in full there, the longest units
    // of a frame, and two workgroups halve them without doubling every other rim segment
    const int kRimSplit = rimSplit == 8 ? 8 : rimSplit == 1 ? 1 : 4;
    const bool rowBorderEight = rimSplit == 48;
    PrefilterPlanHost p;
    const int W = (int)width, H = (int)height;
    p.tilesX = (W + kPTW - 1) / kPTW;
    const int tilesY = (H + kPTH - 1) / kPTH;
    p.tiles = p.tilesX * tilesY;
    p.tileMap.assign((size_t)p.tiles, 0xFFFFFFFFu);
    const int everywhere = (slots > 0 && p.tiles * 2 <= slots) ? std::max(2, std::min(8, slots / std::max(p.tiles, 1))) : 1;
    for (int pass = 0; pass < 2; ++pass) {                 // pass 0: rim tiles, pass 1: interior tiles
        for (int t = 0; t < p.tiles; ++t) {
            const int ty = t / p.tilesX, tx = t - ty * p.tilesX;
            const int bx0 = tx * kPTW - kB / 2, by0 = ty * kPTH - kB / 2;
            const bool rim = !((bx0 >= 0) && (bx0 + kPTW + kB - 2 < W) && (by0 >= 0) && (by0 + kPTH + kB - 2 < H));
            if (rim != (pass == 0)) continue;
            const bool bySegment = rim && everywhere == 1 && kRimSplit > 1;
            const int n = bySegment ? kRimSplit : everywhere;
            if (n == 1) { p.unitMap.push_back((uint32_t)t | (1u << 24)); p.unitAux.push_back(0xFFFFFFFFu); continue; }
            const uint32_t aux0 = (uint32_t)p.auxUnits;
            // (tileMap: first private block | parts << 24 | mask of the segments that have twice as many << 28)
            uint32_t doubled = 0u;
            if (bySegment && rowBorderEight && n == 4) {
                for (int seg = 0; seg < kPTH / kSeg && ty * kPTH + seg * kSeg < H; ++seg) {
                    const int r0 = by0 + kSeg * seg;
                    if (r0 < 0 || r0 + kSegD - 1 >= H) doubled |= 1u << seg;
                }
            }
            p.tileMap[(size_t)t] = aux0 | ((uint32_t)n << 24) | (doubled << 28);
            p.auxUnits += doubled ? 2 * n : n;
            if (bySegment) {
                for (int seg = 0; seg < kPTH / kSeg && ty * kPTH + seg * kSeg < H; ++seg) {
                    const int nSeg = ((doubled >> seg) & 1u) ? 2 * n : n;
                    for (int c0 = 0; c0 < nSeg; c0 += 4) {
                        p.unitMap.push_back((uint32_t)t | ((uint32_t)c0 << 20) | ((uint32_t)nSeg << 24) | (1u << 28) | ((uint32_t)seg << 29));
                        p.unitAux.push_back(aux0);
                    }
                }
            } else {
                for (int c = 0; c < n; ++c) {
                    p.unitMap.push_back((uint32_t)t | ((uint32_t)c << 20) | ((uint32_t)n << 24));
                    p.unitAux.push_back(aux0);
                }
            }
        }
    }
    // Dispatch order.  Workgroups draw units in table order and the device holds only so many at once; the long units are
    // the segments that touch the image border itself (the strip a pan exposes lies there, and so do the rows and columns the
    // upscaler filters differently), so those go first -- all of them start at once -- then the other segments of the rim
    // tiles, then the interior.
    {
        auto touchesBorder = [&](uint32_t um) {
            if (((um >> 28) & 1u) == 0u) return false;
            const int t = (int)(um & 0xFFFFFu), seg = (int)((um >> 29) & 3u);
            const int ty = t / p.tilesX, tx = t - ty * p.tilesX;
            const int bx0 = tx * kPTW - kB / 2, by0 = ty * kPTH - kB / 2 + kSeg * seg;
            return !((bx0 >= 0) && (bx0 + kPTW + kB - 2 < W) && (by0 >= 0) && (by0 + kSegD - 1 < H));
        };
        std::vector<std::pair<uint32_t, uint32_t>> units(p.unitMap.size());
        for (size_t i = 0; i < units.size(); ++i) units[i] = {p.unitMap[i], p.unitAux[i]};
        std::stable_partition(units.begin(), units.end(), [&](const std::pair<uint32_t, uint32_t> &u) { return touchesBorder(u.first); });
        for (size_t i = 0; i < units.size(); ++i) { p.unitMap[i] = units[i].first; p.unitAux[i] = units[i].second; }
    }
    p.units = (int)p.unitMap.size();
    // the whole tiles the lean kernel takes first (motion_lean.hip): their units go to the END of the table -- a call that went
    // through that kernel draws the table without them and takes the tiles it left from the kernel's list instead
    std::vector<uint32_t> lean;
    {
        auto isLean = [&](uint32_t um) { return ((um >> 24) & 0xFu) == 1u && ((um >> 28) & 1u) == 0u && lean_tile_ok((int)(um & 0xFFFFFu), p.tilesX, W, H); };
        std::vector<std::pair<uint32_t, uint32_t>> units(p.unitMap.size());
        for (size_t i = 0; i < units.size(); ++i) units[i] = {p.unitMap[i], p.unitAux[i]};
        std::stable_partition(units.begin(), units.end(), [&](const std::pair<uint32_t, uint32_t> &u) { return !isLean(u.first); });
        for (size_t i = 0; i < units.size(); ++i) { p.unitMap[i] = units[i].first; p.unitAux[i] = units[i].second; }
        for (uint32_t um : p.unitMap) if (isLean(um)) lean.push_back(um & 0xFFFFFu);
    }
    if (LFG_LEAN_XCD_BANDS && lean.size() >= 64) {
        const size_t n = lean.size(), per = (n + 7) / 8;
        for (size_t i = 0; i < 8 * per; ++i) {
            const size_t src = (i % 8) * per + i / 8;
            if (src < n) p.leanTiles.push_back(lean[src]);
        }
    } else {
        p.leanTiles = lean;
    }
    // ... and of the rim tiles above and below them it takes the segments that lie inside the image as the tiles above do
    // (lean_segment_ok): the plan keeps its units for them, which leave at once when they find the segment settled.
    if (LFG_LEAN_PARTIAL && !p.leanTiles.empty()) {
        for (int t = 0; t < p.tiles; ++t) {
            if (lean_tile_ok(t, p.tilesX, W, H)) continue;
            uint32_t mask = 0u;
            for (int seg = 0; seg < kPTH / kSeg; ++seg) mask |= lean_segment_ok(t, seg, p.tilesX, W, H) ? (1u << seg) : 0u;
            // (only the tiles the plan cut into segment units of four parts each: those are the units that look at the marks)
            const uint32_t tm = p.tileMap[(size_t)t];
            if (mask != 0u && tm != 0xFFFFFFFFu && ((tm >> 24) & 0xFu) == 4u) p.leanPartial.push_back((uint32_t)t | ((mask & ~(tm >> 28)) << 24));
        }
        p.leanPartial.erase(std::remove_if(p.leanPartial.begin(), p.leanPartial.end(), [](uint32_t e) { return (e >> 24) == 0u; }), p.leanPartial.end());
    }
    return p;
}

}  // namespace lfg
