// motion_resolve.hip -- the last word of the prefiltered motion path (shaders/motion.comp:33-52): for every pixel the prefilter
// left open, the recorded candidates that pass the final bound (typically one) get the shader's own 64-term chain; the smallest
// (cost, rank in the tie order) wins, which is the shader's first strict minimum in scan order.  See motion_prefilter.hip for
// what is recorded and why no exact minimiser is ever lost ("Bracket").
#include "lfg_motion_common.hpp"

#include <algorithm>

namespace lfg {

LFG_STAMP(
__device__ unsigned long long gResolveStats[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, ~0ull, 0, 0};
void motion_resolve_stats_print() {      // (called by motion_stamps_report, motion_stamps.inc)
    unsigned long long rs[12];
    hipMemcpyFromSymbol(rs, HIP_SYMBOL(gResolveStats), sizeof(rs));
    fprintf(stderr, "resolve (previous calls together): %llu working waves, mean gather %.2f us, mean rest %.2f us, longest wave %.1f us, "
                    "%llu pixels taken together, first start to last end (all calls and what lies between) %.1f us\n",
            rs[4], rs[4] ? rs[5] / 100.0 / rs[4] : 0.0, rs[4] ? rs[6] / 100.0 / rs[4] : 0.0, rs[7] / 100.0, rs[8], (rs[10] - rs[9]) / 100.0);
    fprintf(stderr, "resolve (previous calls together): %llu pixels with several survivors (%llu within 64 px of the rim), %llu survivors\n", rs[0], rs[2], rs[1]);
}
)

// The resolve kernel's launch: a fixed number of waves that share out the rows of the segments the prefilter left open -- the
// compact list its units wrote as they ended (sp.openList, sp.openCount): item i = row i % 16 of open segment i / 16, lane =
// pixel column of the tile (56 of 64 lanes).  Round 2 launched one workgroup
// per 64 x 4 pixels of the frame -- 32,400 at 4K -- of which nine in ten read three flags and left: 51 us under a pan for ~550
// segments with work.  Nothing waits for anything here: the list is complete when this launch starts.
template <bool kFused>
__global__ __launch_bounds__(256) void motion_resolve_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H, const Rec *__restrict__ list,
    const float *__restrict__ uminIn, const uint32_t *__restrict__ countIn, const uint32_t *__restrict__ tileFlags,
    int tilesX, PrefilterPlan sp, const uint32_t *__restrict__ rank2scan, const uint32_t *__restrict__ segDone) {
    __shared__ float sDist[4][kB * kB];    // one block of distances per wave (cooperative exact evaluation)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // How the rows get to the waves -- wave w takes items w, w + waves, ... -- is decided by the GRID (launch_motion_prefiltered_8_16):
    //   * a context that runs one frame at a time launches kResolveGroups (lfg_motion_common.hpp) workgroups, and every row is dealt out: nothing but
    //     these waves runs, they all start at once, and a frame's open rows are in flight together (19 us under a pan);
    //   * a context with frames in flight launches one workgroup per four rows of the WORST case (every segment open), so
    //     each wave has at most one row and the workgroups beyond the list's end leave after one scalar load: the hardware
    //     dispatcher hands the rows to whatever slots the other frames' kernels free -- dealt out statically, a workgroup
    //     that got its slot late held all its rows back (occluded frames 975 -> 915 frames/s with three frames in flight).
    // Drawing rows with counters instead was measured and dropped: device-scope atomics AND loads on one line complete one
    // every 14 ns -- one counter: 1.8 ms on a frame of noise, 56 us before the first wave of a frame with nothing to resolve
    // has learnt that; 32 striped counters with a one-load look at all of them: 5 ms; a draw per segment and workgroup: four
    // rows at a time put four memory latencies in a row (48 instead of 31 us under a pan).
    const uint32_t items = *sp.openCount * (uint32_t)kSeg;
    for (uint32_t item = blockIdx.x * 4u + (uint32_t)wv; item < items; item += gridDim.x * 4u) {
    {
    const int kSegment = (int)sp.openList[item / (uint32_t)kSeg];
    const int px = ((kSegment / (kPTH / kSeg)) % sp.tilesX) * kPTW + lane;
    const int py = ((kSegment / (kPTH / kSeg)) / sp.tilesX) * kPTH + kSeg * (kSegment % (kPTH / kSeg)) + (int)(item % (uint32_t)kSeg);
LFG_STAMP(
    const unsigned long long stampT0 = __builtin_amdgcn_s_memrealtime();
)
    // Candidates are identified by their RANK in the tie order everywhere in the prefiltered path (lists, the
    // zero-cost encoding), so "smallest rank among equal costs" is the tie-break; rank2scan turns it into (dx, dy).
    // No early exits: every lane stays for the cooperative part below.  `live` = this lane owns a pixel to resolve
    // (inside the image and not in a tile that goes through the exact kernel).
    const bool inside = lane < kPTW && px < W && py < H;
    const int cpx = min(px, W - 1), cpy = min(py, H - 1);
    // Everything a pixel of a whole tile usually needs -- threshold, count, first record -- is read up front, next to
    // the two table look-ups and independent of them: one memory latency for the common case instead of a chain of four.
    const size_t pix = (size_t)cpy * (size_t)W + (size_t)cpx;
    const float thr0 = uminIn[pix];
    const uint32_t cnt0 = countIn[pix];
    const Rec rec0 = list[(size_t)cpy * (size_t)kListK * (size_t)W + (size_t)cpx];
    const uint32_t flagged = tileFlags[(cpy / kTH) * tilesX + cpx / kTW];
    const int ptile = (cpy / kPTH) * sp.tilesX + cpx / kPTW;
    const uint32_t tm = sp.tileMap[ptile];
    const uint32_t sm = sp.segMap[ptile * (kPTH / kSeg) + (cpy % kPTH) / kSeg];      // segment handed over at run time?
    // A pixel of a segment the prefilter settled is final (and has no threshold or count: settled segments do not
    // write them); this block got here because the other tile it touches has work left.
    const bool settledSeg = segDone[ptile * (kPTH / kSeg) + (cpy % kPTH) / kSeg] != 0u;
    // ... and so is a pixel the strip kernel has decided (motion_strip.hip): the units that own its segment parked it
    bool stripDecided = false;
    if (sp.colBand != nullptr) {
        const uint32_t cb = sp.colBand[cpy], rb = sp.rowBand[cpx];
        stripDecided = strip_decided(cb, rb, cpx, cpy, W, H);
    }
    const bool live = inside && flagged == 0u && !settledSeg && !stripDecided;
    // Where this pixel's records live: one list in the image-shaped arrays, or several lists (one per unit that
    // shared the tile's candidates) in the auxiliary arrays -- blocks of a tile's 64 rows for the units of the plan,
    // of a segment's 16 rows for a segment handed over at run time.  Record k of list c: recs[c * listStride + k * recStride].
    const bool handedOver = tm == 0xFFFFFFFFu && sm != 0u;
    const bool whole = tm == 0xFFFFFFFFu && sm == 0u;
    // (a tile of the plan: its parts, twice as many for the segments marked in the top four bits)
    const int nLists = whole ? 1 : handedOver ? (int)((sm >> 24) & 0xFu)
                                              : (int)((tm >> 24) & 0xFu) << ((tm >> (28 + (cpy % kPTH) / kSeg)) & 1u);
    const Rec *recs;
    const float *thrs;
    const uint32_t *cnts;
    uint32_t listStride, recStride, thrStride;
    const uint32_t listDepth = handedOver ? (uint32_t)kListDyn : whole ? (uint32_t)kListK : (uint32_t)kListAux;
    if (handedOver) {
        const size_t blk0 = (size_t)(sm & 0xFFFFFFu);
        const int ly = cpy % kSeg, lx = cpx % kPTW;
        recs = sp.dynList + (blk0 * kSeg + (size_t)ly) * (size_t)kListDyn * kPTW + (size_t)lx;
        thrs = sp.dynUmin + (blk0 * kSeg + (size_t)ly) * kPTW + (size_t)lx;
        cnts = sp.dynCount + (blk0 * kSeg + (size_t)ly) * kPTW + (size_t)lx;
        listStride = (uint32_t)(kSeg * kListDyn * kPTW); recStride = kPTW; thrStride = (uint32_t)(kSeg * kPTW);
    } else if (whole) {
        recs = list + (size_t)cpy * (size_t)kListK * (size_t)W + (size_t)cpx;
        thrs = uminIn + (size_t)cpy * (size_t)W + (size_t)cpx;
        cnts = countIn + (size_t)cpy * (size_t)W + (size_t)cpx;
        listStride = 0u; recStride = (uint32_t)W; thrStride = 0u;
    } else {
        const size_t unit0 = (size_t)(tm & 0xFFFFFFu);
        const int ly = cpy % kPTH, lx = cpx % kPTW;
        recs = sp.auxList + (unit0 * kPTH + (size_t)ly) * (size_t)kListAux * kPTW + (size_t)lx;
        thrs = sp.auxUmin + (unit0 * kPTH + (size_t)ly) * kPTW + (size_t)lx;
        cnts = sp.auxCount + (unit0 * kPTH + (size_t)ly) * kPTW + (size_t)lx;
        listStride = (uint32_t)(kPTH * kListAux * kPTW); recStride = kPTW; thrStride = (uint32_t)(kPTH * kPTW);
    }
    // A recorded candidate whose block leaves prev stands for its whole plateau (block_leaves_prev): same exact
    // cost, so the plateau's first member in tie order takes its place.
    auto leaves = [&](int qx, int qy, uint32_t rank) {
        const int scan = (int)rank2scan[rank];
        return block_leaves_prev(qx, qy, scan % kSide - kR, scan / kSide - kR, W, H);
    };
    // Called by the whole wave with the same arguments: the lanes test 64 ranks at a time (a serial scan of up to
    // 1089 table look-ups per pixel made a handful of rim pixels the longest-running part of this kernel).
    auto firstOfPlateau = [&](int qx, int qy, uint32_t rank) {
        if (!leaves(qx, qy, rank)) return rank;
        for (uint32_t r0 = 0; r0 < rank; r0 += 64u) {
            const uint32_t r = r0 + (uint32_t)lane;
            const unsigned long long hit = __ballot(r < rank && leaves(qx, qy, r));
            if (hit != 0ull) return r0 + (uint32_t)__builtin_ctzll(hit);
        }
        return rank;
    };
    float bound = __builtin_inff();        // thresholds are monotone: the tightest one holds for every list
    uint32_t survivors = 0u, bestC = 0u;
    if (live) {
        // Threshold, count and first record of up to 8 lists: all loads of a pixel are issued together (a shared
        // tile's pixel would otherwise walk a chain of a dozen dependent reads).
        constexpr int kMaxLists = 8;
        float thrL[kMaxLists];
        uint32_t cntL[kMaxLists];
        Rec recL[kMaxLists];
        // (branch-free: a list that does not exist reads list 0 and is masked afterwards -- per-list branches made
        //  every list a memory latency of its own)
#pragma unroll
        for (int c = 0; c < kMaxLists; ++c) {
            const size_t cc = (!whole && c < nLists) ? (size_t)c : 0u;
            thrL[c] = thrs[cc * thrStride];
            cntL[c] = cnts[cc * thrStride];
            recL[c] = recs[cc * listStride];
        }
#pragma unroll
        for (int c = 0; c < kMaxLists; ++c) {
            const bool exists = whole ? c == 0 : c < nLists;
            thrL[c] = exists ? thrL[c] : __builtin_inff();
            cntL[c] = exists ? cntL[c] : 0u;
        }
        if (whole) { thrL[0] = thr0; cntL[0] = cnt0; recL[0] = rec0; }
#pragma unroll
        for (int c = 0; c < kMaxLists; ++c) bound = __builtin_fminf(bound, thrL[c]);
        if (bound < 0.5f) {                // a zero-cost candidate exists; the first one in tie order is encoded here
            bestC = __builtin_bit_cast(uint32_t, bound) - 0x00800000u;
        } else {
            // A single survivor IS the shader's answer (the exact minimiser always survives): no evaluation.
            auto count = [&](const Rec rec, bool valid) {
                if (valid && rec_cost_low(rec) <= bound) {
                    if (survivors == 0u) bestC = rec_cand(rec);
                    ++survivors;
                }
            };
            uint32_t nL[kMaxLists];
#pragma unroll
            for (int c = 0; c < kMaxLists; ++c) {
                nL[c] = min(cntL[c], listDepth);
                count(recL[c], nL[c] > 0u);
            }
            // The rest of the lists: four lists x four records per round, all sixteen loads in flight (a pixel
            // without a match holds a handful of records in each list, and one load per loop trip was a memory
            // latency each -- 30 us per wave at the rim).  Survivors are counted in list order within a round and
            // round by round; which of them is remembered as "the" survivor only matters when there is exactly one.
#pragma unroll
            for (int c0 = 0; c0 < kMaxLists; c0 += 4) {
                const uint32_t nMax = max(max(nL[c0], nL[c0 + 1]), max(nL[c0 + 2], nL[c0 + 3]));
                for (uint32_t k0 = 1; k0 < nMax; k0 += 4u) {
                    Rec r[4][4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (uint32_t j = 0; j < 4u; ++j)
                            r[c][j] = recs[(size_t)(c0 + c) * listStride * (nL[c0 + c] > 0u ? 1u : 0u) +
                                           (size_t)min(k0 + j, max(nL[c0 + c], 1u) - 1u) * recStride];
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
#pragma unroll
                        for (uint32_t j = 0; j < 4u; ++j) count(r[c][j], k0 + j < nL[c0 + c]);
                    }
                }
            }
        }
    }
    // Several candidates within the bracket of the minimum (a few percent of the pixels on smooth content): the
    // literal chain of motion.comp:33-47 decides.  The wave takes such pixels one at a time and evaluates each
    // surviving candidate TOGETHER: lane j computes the distance of block position j (row-major, as the shader
    // walks it), the 64 distances go through LDS and every lane adds them in the shader's order -- 64 loads and
    // distances per lane and candidate would otherwise run with one or two lanes active.
    unsigned long long todo = __ballot(survivors > 1u);
LFG_STAMP(
    const unsigned long long stampT1 = __builtin_amdgcn_s_memrealtime();
    const int stampTodo = __builtin_popcountll(todo);
    if (survivors > 1u) { atomicAdd(&gResolveStats[0], 1ull); atomicAdd(&gResolveStats[1], (unsigned long long)survivors);
                          atomicAdd(&gResolveStats[2], (unsigned long long)(px < 64 || py < 64 || px >= W - 64 || py >= H - 64)); }
)
    while (todo != 0ull) {
        const int L = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        const int qx = __shfl(px, L), qy = __shfl(py, L);
        const float qBound = __shfl(bound, L);
        const int qLists = __shfl(nLists, L);
        const uint32_t qDepth = __shfl(listDepth, L);
        const uint32_t qListStride = __shfl(listStride, L), qRecStride = __shfl(recStride, L), qThrStride = __shfl(thrStride, L);
        const Rec *qRecs = reinterpret_cast<const Rec *>(
            ((unsigned long long)__shfl((uint32_t)((unsigned long long)recs >> 32), L) << 32) |
            (unsigned long long)__shfl((uint32_t)(unsigned long long)recs, L));
        const uint32_t *qCnts = reinterpret_cast<const uint32_t *>(
            ((unsigned long long)__shfl((uint32_t)((unsigned long long)cnts >> 32), L) << 32) |
            (unsigned long long)__shfl((uint32_t)(unsigned long long)cnts, L));
        // this lane's block position and its curr texel
        const int cx = qx - kB / 2 + (lane & 7), cy = qy - kB / 2 + (lane >> 3);
        const bool posIn = cx >= 0 && cx < W && cy >= 0 && cy < H;
        const uint32_t ctex = posIn ? *reinterpret_cast<const uint32_t *>(curr + (size_t)cy * (size_t)currPitch + (size_t)cx * 4u) : 0u;
        const float cc[4] = {unorm8_to_float(byte0(ctex)), unorm8_to_float(byte1(ctex)),
                             unorm8_to_float(byte2(ctex)), unorm8_to_float(byte3(ctex))};
        float bestV = __builtin_inff();
        uint32_t bestR = 0xFFFFFFFFu;
        // The pixel's records, GATHERED: the lanes load the lists' counts (one per list), then 64 records per round -- lane j
        // the j-th record of the lists laid end to end -- and the survivors' table entries; the loop below then takes the
        // survivors out of the lanes.  (Round 2 walked list by list and record by record with the same address in every lane:
        // a dependent memory round trip per RECORD, forty in a row for a pixel of a handed-over segment -- the rows that hold
        // many such pixels set this kernel's duration on the occluded and unmatched frames.)
        constexpr int kMaxLists = 8;
        const uint32_t nMine = lane < qLists ? min(qCnts[(size_t)lane * qThrStride], qDepth) : 0u;
        uint32_t nOf[kMaxLists], total = 0u;
#pragma unroll
        for (int c = 0; c < kMaxLists; ++c) { nOf[c] = (uint32_t)__builtin_amdgcn_readlane((int)nMine, c); total += nOf[c]; }
        for (uint32_t base = 0u; base < total; base += 64u) {
            uint32_t k = base + (uint32_t)lane, c = 0u;
            const bool valid = k < total;
#pragma unroll
            for (int cc = 0; cc < kMaxLists - 1; ++cc) {
                const bool beyond = c == (uint32_t)cc && k >= nOf[cc];
                k -= beyond ? nOf[cc] : 0u;
                c += beyond ? 1u : 0u;
            }
            const Rec recMine = valid ? qRecs[(size_t)c * qListStride + (size_t)k * qRecStride] : 0u;
            const bool survives = valid && rec_cost_low(recMine) <= qBound;
            const uint32_t scanMine = survives ? rank2scan[rec_cand(recMine)] : 0u;
            unsigned long long m = __ballot(survives);
            while (m != 0ull) {
                const int bIdx = __builtin_ctzll(m);
                m &= m - 1ull;
                const Rec rec = (Rec)__builtin_amdgcn_readlane((int)recMine, bIdx);
                const int cscan = __builtin_amdgcn_readlane((int)scanMine, bIdx);
                const int dy = cscan / kSide - kR, dx = cscan % kSide - kR;
                const int sx = cx + dx, sy = cy + dy;
                uint32_t ptex = 0u;
                if (posIn && sx >= 0 && sy >= 0 && sx < W && sy < H)
                    ptex = *reinterpret_cast<const uint32_t *>(prev + (size_t)sy * (size_t)prevPitch + (size_t)sx * 4u);
                const f32x4 pp = {unorm8_to_float(byte0(ptex)), unorm8_to_float(byte1(ptex)),
                                  unorm8_to_float(byte2(ptex)), unorm8_to_float(byte3(ptex))};
                // positions outside the image are skipped by the shader; adding +0.0f leaves its sum unchanged
                sDist[wv][lane] = posIn ? dist4<true>(cc, pp) : 0.0f;
                wave_lds_sync();
                float v = 0.0f;
#pragma unroll
                for (int i = 0; i < kB * kB; ++i) v += sDist[wv][i];
                wave_lds_sync();
                const uint32_t r = firstOfPlateau(qx, qy, rec_cand(rec));
                if (v < bestV || (v == bestV && r < bestR)) { bestV = v; bestR = r; }
            }
        }
        if (lane == L) bestC = bestR;
    }
    // single survivors that stand for a plateau: again one pixel at a time, the wave searching together
    unsigned long long stands = __ballot(live && survivors == 1u && leaves(px, py, bestC));
    while (stands != 0ull) {
        const int L = __builtin_ctzll(stands);
        stands &= stands - 1ull;
        const uint32_t r = firstOfPlateau(__shfl(px, L), __shfl(py, L), (uint32_t)__shfl((int)bestC, L));
        if (lane == L) bestC = r;
    }
    if (live) {
        const int bscan = (int)rank2scan[bestC];
        const int dyi = bscan / kSide, dxi = bscan - dyi * kSide;
        if (kFused && sp.fused.data) fused_pixel(sp.fused, prev, prevPitch, curr, currPitch, W, H, px, py, dxi - kR, dyi - kR);
        int8_t *dst = mv + (size_t)py * (size_t)mvPitch + (size_t)px * 2u;
        if (!kFused || sp.fused.storeMv)
        *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(uint8_t)(int8_t)(dxi - kR) | (uint16_t)((uint16_t)(uint8_t)(int8_t)(dyi - kR) << 8);   // both components, one store
    }
LFG_STAMP(
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&gResolveStats[4], 1ull);                     // working waves
        atomicAdd(&gResolveStats[5], stampT1 - stampT0);        // gather time (10 ns ticks)
        atomicAdd(&gResolveStats[6], t2 - stampT1);             // cooperative + plateau + write time
        atomicMax(&gResolveStats[7], t2 - stampT0);             // longest wave
        atomicAdd(&gResolveStats[8], (unsigned long long)stampTodo);
        atomicMin(&gResolveStats[9], stampT0); atomicMax(&gResolveStats[10], t2);
    }
)
    }
    }   // items
}

hipError_t launch_motion_resolve(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv, const uint32_t *list,
                                 const float *umin, const uint32_t *count, const uint32_t *tileFlags, int tilesX, const PrefilterPlan &sp,
                                 const uint32_t *rank2scan, const uint32_t *segDone, int groups) {
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)std::max(1, groups)), dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height, (const Rec *)list, umin, count, tileFlags, tilesX, sp, rank2scan, segDone);
    };
    if (sp.fused.data) launch(motion_resolve_kernel<true>); else launch(motion_resolve_kernel<false>);
    return hipGetLastError();
}

}  // namespace lfg
