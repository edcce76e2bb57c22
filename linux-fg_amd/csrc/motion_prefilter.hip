// motion_prefilter.hip -- the prefiltered path of shaders/motion.comp:16-57 (blockSize 8, searchRadius 16, as dispatched by
// src/frame_manager.cpp:325-344): the persistent kernel and the launch sequence of one lfg_motion call.  The literal kernel
// it is checked against, and that flagged tiles go through: motion_literal.hip; the file map: lfg_motion_common.hpp.
//
// One work unit (prefilter_unit) is a single function of ~2,000 lines whose phases share two dozen wave-private values
// (thresholds, counts, the lane's 23 texels ...) as lambdas over common state: it is kept in one scope -- separately compiled
// device functions would pass that state through memory -- and split BY PHASE into the prefilter_*.inc files it includes, in
// the order a unit runs them:
//     prefilter_staging.inc   the search window into LDS
//     prefilter_sums.inc      column sums, slab, thresholds / counts / lists of the wave's pixels, row sums
//     prefilter_records.inc   the test of a fully evaluated candidate against the pixels' thresholds, and the record path
//     prefilter_tests.inc     (inside run()) zero bound, squared distance, exact-match test of the lattice, the band
//     prefilter_walks.inc     four-point, one-point (lattice) and eight- / sixteen-point tests of a batch of 64 candidates
//     prefilter_flat.inc      a window of one colour
//     prefilter_narrow.inc    narrow search: a band of at most sixteen unmatched pixel columns
//     prefilter_rowband.inc   row band: at most eight unmatched pixel rows
//     prefilter_batches.inc   the batch loop: hints, hand-over, lookahead, by rank, deferred survivors, full evaluations
//     prefilter_epilogue.inc  flags, thresholds, settling in place, pooling the four parts of a segment unit
//
// The exact chain of motion.comp costs 64 dependent-order adds per (pixel, candidate) plus a correctly
// rounded distance per (position, candidate), but only the candidates that can still be THE minimum need
// that.  Everything else is ruled out by a cheap value that provably brackets the shader's cost.
//
// Bracket.  Let T = sum over the block of sqrt(n_c)/255 in real arithmetic, n_c = sum over the four channels
// of (curr byte - prev byte)^2, an integer <= 4*255^2.  With u = 2^-24:
//   * the shader's fp32 cost S: a texel is fl(k/255) (relative error <= u), so a channel difference with
//     byte difference D != 0 is (D/255)(1+e), |e| <= (ka+kb)u/|D| + u <= 510u (D = 0 gives exactly 0); squares,
//     the three adds and the correctly rounded sqrt turn that into a distance within 516u of sqrt(n_c)/255,
//     and the 63 sequential adds of non-negative terms add at most 63u:      |S/T - 1| <= 580u = 3.5e-5;
//   * the prefilter's S~: n_c exactly (integer dot products), one v_sqrt_f32 (1 ulp <= 2u), a depth-6
//     pairwise tree (6u):                                                    |S~/(255 T) - 1| <= 9u.
// Hence S lies within a factor (1 -/+ 3.6e-5) of S~/255 whatever the data, and a candidate m can only be the
// exact minimum while  S~(m) <= kRatio * min_j S~(j),  kRatio = 1 + 8e-5 >= (1+3.6e-5)/(1-3.6e-5) with room
// for the rounding of the product.
//
//   motion_hint_kernel, motion_order_kernel   this call's visiting order: the SAD-best candidates of 256 sample blocks in
//       front (most popular first), then a fixed pseudo-random order of the rest; the hint kernel also clears the call's
//       control area.
//   motion_prefilter_kernel  a wave owns a 16-row segment of a 56 x 64 tile.  Per BATCH of up to 64 candidates (one
//       per lane) a partial-distortion test drops every candidate whose distances at a lattice of block positions
//       already exceed all of the segment's thresholds (one point per block, then 2 x 2 and 4 x 4 sums, over the column
//       band of the pixels that are not settled yet; exact-texel compares once a wave is settled; see run()).  The
//       survivors are evaluated in full: per-position distances -> 8-row column sums (registers) -> 8-column row
//       sums = S~ per pixel; a candidate is recorded in the pixel's list when S~ <= kRatio * (min of S~ over the
//       candidates seen so far) (it may still be the exact minimum); at most kListK / kListAux / kListDyn per pixel, else the tile is
//       flagged and left to the exact kernel.  Work units (prefilter_plan): whole tiles; rim tiles as one
//       workgroup per segment with its four waves on quarters of the order; segments of whole tiles that find no
//       match are handed over through a queue that the persistent workgroups of the SAME launch drain, in eight parts
//       that start from the thresholds of the wave that handed them over.  Easy pixels are settled in the kernel itself.
//   motion_resolve_kernel    per remaining pixel: the recorded candidates that pass the final bound (typically
//       one) get the literal chain of motion.comp:33-47; the smallest (cost, rank in the tie order) wins, which
//       is the shader's first strict minimum in scan order.
//   motion_tiled_8_16_kernel with the tile flags: full exact evaluation of the flagged tiles (flat or
//       finely tied content, where every candidate survives the filter).
// Every exact minimiser m* is recorded and survives: S~(m*)(1-3.6e-5) <= 255 S(m*) <= 255 S(j) <=
// S~(j)(1+3.6e-5) for all j.  Results are therefore identical to the exact kernel's for any input; only the run time depends
// on the content.
//
// Data flow of the prefilter (no texture unit, no fp32 texels -- bytes and integer dot products):
//   * a 256-thread workgroup owns a 56 x 64 pixel tile = 63 x 71 block positions; prev's search window
//     (95 x 103 texels, zero outside the image like texelFetch) is staged ONCE in LDS as packed RGBA8;
//     after that its four waves never synchronise again: wave `seg` owns pixel rows 16 seg .. 16 seg + 15;
//   * column sums, thread = position column `lane`: the 23 curr texels of rows 16 seg .. 16 seg + 22 stay in
//     registers (packed, plus their squared norms); per candidate 23 conflict-free ds_read_b32 of the
//     window, n = |c|^2 + |p|^2 - 2 c.p by two v_dot4_u32_u8 whose accumulators are float bit patterns (so n
//     comes out as a float without conversions), v_sqrt_f32, then the 16 column sums
//     V8(r) = D(r) + ... + D(r+7) as a shared pairwise tree (58 adds);
//   * transposition through a wave-private LDS slab, all sixteen rows in one round trip, rows r and r + 8 interleaved
//     (writes: lane = column, one ds_write_b64 per row pair; reads: lane = (row lane&7, 7-pixel run lane>>3), one
//     ds_read_b64 per input column: both of the lane's rows at once);
//   * row sums, per lane two runs of 7 pixels (rows r and r+8) packed side by side: 14 input pairs -> shared
//     pairwise tree (31 packed adds) -> 7 + 7 S~, threshold test, list append.
//   LDS: 38.2 KB window + 4 x 5.1 KB slabs (the wide passes use 4.3 KB of each) + 4.3 KB visiting order + 2.2 KB its
//   inverse + 8 KB narrow-search state + 2 KB waiting candidates = 75 KB; 256 VGPRs -> two workgroups (8 waves) per CU.
//   DS operations of one wave execute in order, so a slab needs neither double buffering nor barriers.  Loop order
//   per surviving candidate e: window reads(e+1) issued | row sums + test(e) | column sums(e+1) | slab
//   write/read(e+1); the slab round trip is the one exposed latency, covered by the other wave of the SIMD.
#include "lfg_motion_common.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace lfg {

constexpr int kSlabP = 132;                       // slab pitch of a ROW PAIR (floats): rows r, r + 8 interleaved per column,
                                                  // so a thread writes two rows with one ds_write_b64 and the row sums read
                                                  // both with one ds_read_b64.  8-byte accesses are served sixteen lanes at a
                                                  // time over the sixteen 8-byte slots of the 32 banks: with 66 slots per row
                                                  // the sixteen lanes (r8 = 0..7, q = 2k, 2k + 1) of a transposed read fall
                                                  // into slots 2 r8 + 7 q + i mod 16 -- the even ones for one q, the odd ones
                                                  // for the other.  (136 put rows r and r + 4 on the same slots: 5.7e8
                                                  // conflict cycles per launch on a frame that searches in full, measured.)
static_assert((kSlabP / 2) % 16 == 2 && kSlabP / 2 >= 64, "conflict-free transposed 8-byte reads");
// Narrow search (prefilter_unit, "Narrow search"): the band of pixel columns that holds a segment's pixels without a
// match, at most sixteen columns wide, searched several candidates per pass.
constexpr int kNarrowMax = 16;                    // pixel columns of the band at most
constexpr int kNarrowPitch = 164;                 // slab pitch of a row pair in the narrow passes (floats): = 4 mod 32, so the
                                                  // transposed reads of 16 rows x 2 column groups fall into 32 distinct banks
constexpr int kNarrowQ = 20;                      // ... whose columns are stored four-way interleaved (column c at (c & 3) * 20 + c / 4)
constexpr int kSlabFloats = 8 * kNarrowPitch;     // a wave's slab: 8 row pairs x kSlabP (wide), 8 row pairs x kNarrowPitch (narrow)
static_assert(kSlabFloats >= 8 * kSlabP && 2 * (3 * kNarrowQ + 15) + 1 < kNarrowPitch, "both layouts fit");

// order32[e] = the candidate's rank in the tie order (motion_tables) in the low half, its window offset
// (dx+R)*kWinH + (dy+R) in the high half; 32-bit entries so the wave-uniform reads are scalar loads.
static_assert(LFG_DYN_PARTS == 4 || LFG_DYN_PARTS == 8, "four parts per queue entry, at most eight lists per pixel in the resolve kernel");

LFG_STAMP(__device__ unsigned long long gMotionStamps[8192 * 4 * 8];)      // (stamps builds: eight words per wave of a unit; motion_stamps.inc reads them)

// One work unit of the prefilter (see motion_prefilter_kernel below, which hands units to its workgroups).
// `unit` indexes the plan's unit table, or -- fromQueue -- the queue of segments handed over at run time.
template <bool kFused, int kTier>
__device__ __forceinline__ void prefilter_unit(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int W, int H, Rec *__restrict__ list, float *__restrict__ uminOut,
    uint32_t *__restrict__ countOut, uint32_t *__restrict__ tileFlags, int flagTilesX,
    const uint32_t *__restrict__ order32, const PrefilterPlan &sp,
    int8_t *__restrict__ mv, int mvPitch, const uint32_t *__restrict__ rank2scan, uint32_t *__restrict__ segDone,
    const int unit, const bool fromQueue, const uint32_t um,
    uint32_t *sWin, float (*sSlab)[kSlabFloats], uint32_t *sOrder, const uint16_t *sInv, uint32_t &sGiveUp, uint32_t (*sNarrow)[2 * kSeg * kNarrowMax],
    uint32_t (*sPending)[128]) {

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave index 0..3
LFG_STAMP(
    const unsigned long long stampStart = __builtin_amdgcn_s_memrealtime();
    unsigned stampEvals = 0u, stampBatches = 0u, stampBox = 0u, stampBand = 0u, stampNarrow = 0u, stampThr = 0u, stampThrEnd = 0u, stampFour = 0u;
    unsigned long long stampStaged = 0ull, stampFirst = 0ull;
    unsigned long long phaseLattice = 0ull, phaseSixteen = 0ull, phaseEval = 0ull;     // (LFG_STAMP_PHASES: time in the batch tests, the sixteen-point test, full evaluations)
)
    // Work units (PrefilterPlan).  A tile is 4 segments of 16 rows, and a workgroup is either
    //   * a whole tile: wave = segment, the entire candidate order;
    //   * one of nChunks contiguous parts of a tile's candidate order: wave = segment, private lists;
    //   * one segment of a tile: wave = one of 4 consecutive parts of the order (of nChunks), private lists --
    //     for tiles whose segments differ widely in cost, so that the waves of a workgroup finish together.
    // `seg` is the segment this wave works on, `wave` its place in the workgroup (its slab).
    //   A whole tile's wave that still has a threshold of 510 or more after the first eight candidates -- its segment
    //   holds pixels without a match, the partial-distortion test cannot fire, all 1089 candidates await a full
    //   evaluation -- hands the segment over instead: it pushes a segment unit onto a queue and returns, and the
    //   workgroups that have run out of plan units take the queued segments, four waves each (motion_prefilter_kernel).
    const bool segUnit = ((um >> 28) & 1u) != 0u;
    const int tile = (int)(um & 0xFFFFFu), nChunks = (int)((um >> 24) & 0xFu);
    const int seg = segUnit ? (int)((um >> 29) & 3u) : wave;
    const int chunk = (int)((um >> 20) & 0xFu) + (segUnit ? wave : 0);
    const bool whole = nChunks == 1;
    const int perChunk = (kCand + nChunks - 1) / nChunks;
    const int eBegin = chunk * perChunk, eEnd = min(kCand, eBegin + perChunk);
    const int tileY = tile / sp.tilesX, tileX = tile - tileY * sp.tilesX;
    const int tx0 = tileX * kPTW, ty0 = tileY * kPTH;                  // tile origin (pixels)
    const int bx0 = tx0 - kB / 2, by0 = ty0 - kB / 2;                  // image coords of block position (0,0)
    // Segments the lean kernel has settled already (motion_lean.hip; whole tiles only): a tile whose four are done is skipped
    // before anything is staged, a done segment's wave leaves behind the staging barrier.
    if (whole && !fromQueue) {
        const uint32_t *const d = segDone + tile * (kPTH / kSeg);
        if (__builtin_amdgcn_readfirstlane((int)(d[0] & d[1] & d[2] & d[3])) != 0) return;
    }
    // ... and so is a segment unit whose segment that kernel has settled (a rim tile's segments inside the image: all four waves
    // of the unit work on the one segment, and leave together)
    if (segUnit && !fromQueue && __builtin_amdgcn_readfirstlane((int)segDone[tile * (kPTH / kSeg) + (int)((um >> 29) & 3u)]) != 0) return;
    if (tid == 0) sGiveUp = 0u;
    // (the two flags of this call's order, requested here: read where they are used -- behind the staging barrier -- each
    //  was a scalar load from memory with nothing to hide its latency, two microseconds per unit)
    const uint32_t orderHandOver = order32[kCand], orderHints = order32[kCand + 1];
    // (the shaders' own tie order: a candidate's rank IS its place in the scan -- rank 0 is the scan's first vector, not the
    //  zero vector the intended order starts with)
    const bool rankIsScan = LFG_RANK_ARITH && rank2scan[0] == 0u;
    const uint32_t unitAuxFirst = (!whole && !fromQueue) ? sp.unitAux[unit] : 0u;      // (likewise: its lists are addressed behind the barrier)
    // A unit that shares its tile first runs the head of the order -- this call's top hint and zero motion -- for the
    // thresholds only (not recorded: the unit that owns those entries records them).  Without it a unit whose part
    // of the order holds no good candidate starts from nothing and records far more than it needs to.
    // (TWO entries.  Eight until round 4 -- zero motion and the first seven hints: where a segment's pixels have no match,
    //  the strips a pan exposes, no test can drop a candidate and each of the eight was a full evaluation, 13 us in front of
    //  every part of every rim segment: 877 units a call.  What the six further hints bought the segments that do match was
    //  less: 2 -> pan 3,415 -> 3,562 frames/s with three frames in flight, 2,640 -> 2,728 one at a time, noisy +1.3 %,
    //  moving objects +0.5 %, occluded +0.2 %, fade +1 %; 3 and 4 lie in between.  One -- the top hint alone -- is the first
    //  batch itself and leaves the second one empty.)
    constexpr int kHead = LFG_HEAD;
    static_assert(kHead > LFG_FIRST_BATCH, "the head is the first batch and at least one entry more");
    // (the parts of a segment handed over at run time start from the thresholds of the wave that handed it over --
    //  which had tried every hint -- instead: see the push and `run` below)
    const int nHead = (chunk > 0 && !(fromQueue && LFG_QUEUE_INIT)) ? kHead : 0;

    // Interior tiles: every block position the outputs use (columns 0..62) lies inside the image.
    const int borderTile = __builtin_amdgcn_readfirstlane(
        !((bx0 >= 0) && (bx0 + kPTW + kB - 2 < W) && (by0 >= 0) && (by0 + kPTH + kB - 2 < H)));
    // ... and what counts for a wave is its own segment's 23 position rows: the inner segments of a tile at the top or bottom
    // border take the interior paths (14-point lattice, lookahead, no validity masks).
    const int segRow0 = __builtin_amdgcn_readfirstlane(by0 + kSeg * seg);
    const int borderSeg = LFG_BORDER_PER_SEGMENT
        ? __builtin_amdgcn_readfirstlane((int)((bx0 < 0) | (bx0 + kPTW + kB - 2 >= W) | (segRow0 < 0) | (segRow0 + kSegD - 1 >= H))) : borderTile;

    // ---- this thread's 23 block positions (column bx0 + lane, rows by0 + 16 seg + j).  Loaded BEFORE the window is staged
    // (their squares are taken after it): the two sets of loads are in flight together, and a unit pays one memory
    // latency before its barrier, not two.
    uint32_t c[kSegD], cc[kSegD];
    uint32_t valid = 0u;                                               // bit j: position j lies inside the image
    {
        // branch-free: the load goes to the nearest texel inside the image and is dropped afterwards (a branch per
        // position is thirty instructions of masks and address arithmetic, 5 KB of code)
        const int gx = bx0 + lane;
        const bool okx = gx >= 0 && gx < W;
        const uint8_t *const column = curr + (size_t)clampi(gx, 0, W - 1) * 4u;
        // (the rows inside the image are one run jLo .. jHi - 1, the same for every lane: a scalar mask, not 23 compares,
        //  selects and shifts per lane -- a tenth of a unit's instructions before its first batch)
        const int gy0 = by0 + kSeg * seg;
        const int jLo = clampi(-gy0, 0, kSegD), jHi = clampi(H - gy0, 0, kSegD);
        const uint32_t rowMask = jHi > jLo ? (((1u << jHi) - 1u) & ~((1u << jLo) - 1u)) : 0u;
        valid = okx ? rowMask : 0u;
        const uint32_t keepLane = okx ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int j = 0; j < kSegD; ++j) {
            const int gy = gy0 + j;
            c[j] = *reinterpret_cast<const uint32_t *>(column + (size_t)clampi(gy, 0, H - 1) * (size_t)currPitch);
        }
        if (borderSeg) {                   // (wave-uniform; a segment inside the image keeps every texel)
#pragma unroll
            for (int j = 0; j < kSegD; ++j) {
                const uint32_t keepRow = 0u - ((rowMask >> j) & 1u);    // scalar: all ones or zero
                c[j] = c[j] & keepLane & keepRow;
            }
        }
    }
#include "prefilter_staging.inc"
    // Some candidate's block can leave prev altogether only if the search window does.
    const int windowLeavesPrev = __builtin_amdgcn_readfirstlane(
        !((bx0 - kR >= 0) && (bx0 + kPTW + kB - 2 + kR < W) && (by0 - kR >= 0) && (by0 + kPTH + kB - 2 + kR < H)));

    __syncthreads();                       // window staged; the only workgroup barrier
LFG_STAMP(
    stampStaged = __builtin_amdgcn_s_memrealtime();
)
    if (ty0 + kSeg * seg >= H) return;     // this wave's rows lie below the image
    if (whole && !fromQueue && __builtin_amdgcn_readfirstlane((int)segDone[tile * (kPTH / kSeg) + seg]) != 0) return;     // (settled by the lean kernel)

    // (lane 63 has no position column: it re-reads lane 62's texels, its sums are never used)
    const lds_ro_u32_ptr winBase = (lds_ro_u32_ptr)(sWin + min(lane, kPTW + kB - 2) * kWinH + kSeg * seg);
#include "prefilter_sums.inc"
#include "prefilter_records.inc"

    bool settledAtZero = false;            // set by run(): the wave's largest threshold stands for a zero cost
    auto run = [&]() -> int {              // 0: done, 1: lists overflowed (tile flagged), 2: segment handed over
#include "prefilter_tests.inc"
#include "prefilter_walks.inc"
#include "prefilter_flat.inc"
#include "prefilter_narrow.inc"
#include "prefilter_rowband.inc"
#include "prefilter_batches.inc"
    };
#include "prefilter_epilogue.inc"
}

// The prefilter's launch: as many workgroups as the device holds at once (prefilter_slots), each taking work units
// until none are left -- first the units of the plan (ctrl[0]: next index; rim segment units come first in the
// table, so the long ones start at once), then the segments that whole tiles hand over AT RUN TIME (ctrl[2]: next
// queue slot; sp.queueCount[0]: slots pushed so far).  A segment is handed over by a wave that finds no match for it
// once every hint has been tried: one such segment used to hold its whole tile's workgroup for milliseconds, and a
// second launch for the queue could only start when the first had drained.  Now the workgroups that run out of
// plan units pick the queued segments up while the long units are still running.
//   ctrl[1] counts finished plan units: only a running plan unit can push, so "all plan units finished and my slot
//   is still empty" ends a workgroup.  A workgroup that waits polls with read-modify-write atomics (the counters
//   and the queue entries are written by atomics on other XCDs; the per-XCD L2s are not coherent for plain
//   accesses) and sleeps ~14 us between polls.  Every wait is bounded by the running units, every workgroup of
//   the grid is resident or finds nothing to wait for: no workgroup ever waits for one that has not started.
// kTier: 0 = the kernel every call takes unless the lane's previous call says otherwise; 1 = the same with the eight-point walk by SADs
// for thresholds of 300 - 768 (prefilter_walks.inc): sensor noise of +-3 .. +-6 levels at the input.  A kernel of its own and not a
// run-time branch: round 4 measured the branch (noise +-4: +18 %) and dropped it because the second copy of a walk inside the one
// kernel moved the register allocation of every other path (+-2: -3.4 %, +-8: -4.7 %, occlusions -1.1 %).
template <bool kFused, int kTier = 0>
__global__ __launch_bounds__(kPNT, LFG_PREF_OCC) void motion_prefilter_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int W, int H, Rec *__restrict__ list, float *__restrict__ uminOut,
    uint32_t *__restrict__ countOut, uint32_t *__restrict__ tileFlags, int flagTilesX,
    const uint32_t *__restrict__ order32, PrefilterPlan sp,
    int8_t *__restrict__ mv, int mvPitch, const uint32_t *__restrict__ rank2scan, uint32_t *__restrict__ segDone,
    uint32_t *__restrict__ ctrl) {
    __shared__ uint32_t sWin[kWinH * kWinW];                           // 38.2 KB packed RGBA8 search window
    __shared__ __attribute__((aligned(8))) float sSlab[kPNT / 64][kSlabFloats];     // 4 x 5.1 KB (the wide passes use 2.1 KB of each)
    __shared__ uint32_t sNarrow[kPNT / 64][2 * kSeg * kNarrowMax];     // 4 x 2 KB: thresholds and counts of a narrow band's pixels
    __shared__ uint32_t sPending[kPNT / 64][128];                      // 4 x 0.5 KB: candidates waiting for the sixteen-point test
    __shared__ uint32_t sOrder[kCand + 7];                             // the visiting order
    __shared__ uint32_t sGiveUp;
    __shared__ uint32_t sNext[2];                                      // {unit | fromQueue << 31, its table entry}
    __shared__ uint16_t sInv[kCand + 1];                               // its inverse: where the order visits rank r
    for (int i = threadIdx.x; i < kCand; i += kPNT) {                  // once: the same for every unit
        const uint32_t o = order32[i];
        sOrder[i] = o;
        sInv[min(o & 0xFFFFu, (uint32_t)kCand)] = (uint16_t)i;
    }
    // Thread 0 hands out the work.  Plan units: ctrl[0] is the next index (fetch-and-add).  Queue slots: ctrl[2] is the
    // next slot, also fetch-and-add -- a compare-and-swap loop collapses when hundreds of workgroups run out of plan
    // units together -- so a workgroup can come to OWN a slot that no segment has been pushed into yet.  It keeps the
    // slot (`owned`) through whatever else it does and does not leave while somebody could still fill it.
    // the units of this launch: the plan's table -- or, behind the lean kernel, the table without that kernel's tiles plus the tiles
    // in which it left a segment (its list is complete: that kernel has finished)
    const uint32_t hardUnits = sp.hardCount ? *sp.hardCount : 0u;
    const uint32_t planUnits = sp.hardCount ? (uint32_t)sp.unitsStatic + hardUnits : (uint32_t)sp.units;
    uint32_t owned = kNoUnit;
    for (;;) {
        __syncthreads();                   // the previous unit is over for all four waves: its LDS may be reused
        if (threadIdx.x == 0) {
            uint32_t next = kNoUnit, entry = 0u;
            const uint32_t cap = (uint32_t)sp.queueCap;
            auto wait_entry = [&](uint32_t h) {           // slot h has been pushed (h < tail): its entry is on its way
                uint32_t e, n = 0u;
                while ((e = (++n & 31u) ? peek(&sp.queue[h]) : peek_hard(&sp.queue[h])) == 0u) __builtin_amdgcn_s_sleep(8);
                return e;
            };
            if (owned != kNoUnit && (entry = peek(&sp.queue[owned])) != 0u) { next = 0x80000000u | owned; owned = kNoUnit; }
            // Queued segments before plan units: they are the long units (a segment that searches in full), and the
            // earlier they start the more of them run next to the plan's units instead of after them.
            if (LFG_QUEUE_FIRST && next == kNoUnit && owned == kNoUnit && peek(&ctrl[kCtrlNextSlot]) < min(peek(sp.queueCount), cap)) {
                const uint32_t h = atomicAdd(&ctrl[kCtrlNextSlot], 1u);
                if (h < min(peek_hard(sp.queueCount), cap)) { entry = wait_entry(h); next = 0x80000000u | h; }
                else if (h < cap) owned = h;              // lost the race for the last filled slot: h is mine when it fills
            }
            if (next == kNoUnit) {        // (drawn when needed, not ahead: a workgroup holding two of the long units in a row
                                          //  would run them one after the other while others idle)
                const uint32_t u = atomicAdd(&ctrl[kCtrlNextUnit], 1u);
                // (the tiles the lean kernel left come FIRST: they hold the segments that will be handed over or searched in full --
                //  an occlusion, a moving object's rim -- and drawn last they were the launch's tail: occluded frames -4 %, moving objects -7 %)
                // (`next` stays the unit's index in the table -- its private lists are found by it; a left tile, whole, has none)
                if (u < planUnits) { next = u < hardUnits ? (uint32_t)sp.unitsStatic + u : u - hardUnits; entry = u < hardUnits ? (sp.hardTiles[u] | (1u << 24)) : sp.unitMap[u - hardUnits]; }
            }
            if (next == kNoUnit) {
                // Out of plan units: wait for my slot to be filled, or for the last plan unit to finish (only a running
                // plan unit can push; its pushes precede its count in ctrl[1]).
                // (A frame in which no segment has been handed over by now -- a clean pan -- is unlikely to start: the
                //  workgroup leaves at once, because even a waiting workgroup costs the long units that still run beside
                //  it a few percent.  Should a late unit push after all, the workgroups still running, in the end the
                //  pusher itself, take the segment.)
                if (owned == kNoUnit && peek_hard(sp.queueCount) != 0u) { const uint32_t h = atomicAdd(&ctrl[kCtrlNextSlot], 1u); if (h < cap) owned = h; }
                uint32_t n = 0u;
                while (owned != kNoUnit) {
                    const bool hard = (++n & 31u) == 0u;
                    if ((entry = hard ? peek_hard(&sp.queue[owned]) : peek(&sp.queue[owned])) != 0u) { next = 0x80000000u | owned; owned = kNoUnit; break; }
                    if ((hard ? peek_hard(&ctrl[kCtrlUnitsDone]) : peek(&ctrl[kCtrlUnitsDone])) >= planUnits) {
                        if (owned < min(peek_hard(sp.queueCount), cap)) { entry = wait_entry(owned); next = 0x80000000u | owned; }
                        owned = kNoUnit;
                        break;
                    }
                    for (int k = 0; k < LFG_PREF_POLL_SLEEPS; ++k) __builtin_amdgcn_s_sleep(127);
                }
            }
            sNext[0] = next; sNext[1] = entry;
        }
        __syncthreads();
        const uint32_t next = sNext[0], entry = sNext[1];
        if (next == kNoUnit) return;
        const bool fromQueue = (next >> 31) != 0u;
        prefilter_unit<kFused, kTier>(prev, prevPitch, curr, currPitch, W, H, list, uminOut, countOut, tileFlags, flagTilesX, order32, sp,
                       mv, mvPitch, rank2scan, segDone, (int)(next & 0x7FFFFFFFu), fromQueue, entry, sWin, sSlab, sOrder, sInv, sGiveUp, sNarrow, sPending);
        if (!fromQueue) {
            __syncthreads();               // every wave of the unit is past its pushes
            if (threadIdx.x == 0) atomicAdd(&ctrl[kCtrlUnitsDone], 1u);
        }
    }
}

#ifdef LFG_MOTION_STAMPS
#include "motion_stamps.inc"
#endif

int prefilter_slots() {
    int dev = 0, cus = 0, perCu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, motion_prefilter_kernel<false>, kPNT, 0) != hipSuccess) return 0;
    return cus * perCu;
}

hipError_t launch_motion_prefiltered_8_16(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                          const lfg_frame &mv, uint8_t *workspace, const MotionWorkspaceLayout &l, int units,
                                          const uint32_t *rank2scan, const uint32_t *order,
                                          const uint32_t *entryOfScan, const uint32_t *baseScan, bool useHints, bool framesInFlight,
                                          const FusedOut &fused, bool lean, uint32_t *leanFlagHost, int groupsCap, bool expectNoFallback,
                                          const MotionKnobs &knobs, bool rankIsScan, int tier) {
    const int tilesX = ((int)curr.width + kTW - 1) / kTW;
    lean = lean && useHints && !fused.data && l.units2 > 0 && l.leanCount > 0 && curr.width >= 64u && curr.height >= 64u && lean_frames_ok(prev, curr, mv);
    if (lean) units = l.units2;          // the plan that goes with the lean kernel
    Rec *list = reinterpret_cast<Rec *>(workspace + l.list);
    float *umin = reinterpret_cast<float *>(workspace + l.umin);
    uint32_t *count = reinterpret_cast<uint32_t *>(workspace + l.count);
    uint32_t *flags = reinterpret_cast<uint32_t *>(workspace + l.tileFlags);
    static_assert(kPTH == kTH, "prefilter tiles and exact tiles share their rows");
    uint32_t *const ctrl = reinterpret_cast<uint32_t *>(workspace + l.ctrl);
    PrefilterPlan sp{};
    sp.tilesX = ((int)curr.width + kPTW - 1) / kPTW;
    sp.units = units;
    sp.unitMap = reinterpret_cast<const uint32_t *>(workspace + (lean ? l.plan2 : l.plan));
    sp.unitAux = sp.unitMap + units;
    sp.tileMap = sp.unitAux + units;
    sp.auxList = reinterpret_cast<uint32_t *>(workspace + l.auxList);
    sp.auxUmin = reinterpret_cast<float *>(workspace + l.auxUmin);
    sp.auxCount = reinterpret_cast<uint32_t *>(workspace + l.auxCount);
    sp.segMap = reinterpret_cast<uint32_t *>(workspace + l.segMap);
    sp.queueCount = reinterpret_cast<uint32_t *>(workspace + l.queueCount);
    sp.queue = reinterpret_cast<uint32_t *>(workspace + l.queue);
    sp.queueCap = l.queueCap;
    sp.dynList = reinterpret_cast<uint32_t *>(workspace + l.dynList);
    sp.dynUmin = reinterpret_cast<float *>(workspace + l.dynUmin);
    sp.dynCount = reinterpret_cast<uint32_t *>(workspace + l.dynCount);
    sp.dynInit = reinterpret_cast<uint32_t *>(workspace + l.dynInit);
    sp.openList = reinterpret_cast<uint32_t *>(workspace + l.openList);
    sp.openCount = ctrl + kCtrlOpenCount;
    sp.colBand = sp.rowBand = nullptr;
    sp.fused = fused;
    sp.dynParts = framesInFlight ? 4 : LFG_DYN_PARTS;
    if (knobs.dynParts) sp.dynParts = knobs.dynParts == 4 ? 4 : LFG_DYN_PARTS;      // (measurement)
    // (as many SEGMENTS as the queue holds in LFG_DYN_PARTS parts, whatever the parts of this call: one block of thresholds each)
    sp.queueCap = l.queueCap / ((LFG_DYN_PARTS / 4) / (sp.dynParts / 4));
    sp.unitsStatic = lean ? l.units2Static : units;
    sp.hardTiles = lean ? reinterpret_cast<const uint32_t *>(workspace + l.hardTiles) : nullptr;
    sp.hardCount = lean ? ctrl + kCtrlHardCount : nullptr;          // (cleared by the hint kernel with the rest)
    uint32_t *segDone = reinterpret_cast<uint32_t *>(workspace + l.segDone);
    hipError_t e = hipSuccess;
    uint32_t *verdictWord = nullptr;       // this call's verdict word (its own order table's entry kCand + 2), where there is one
    if (useHints && curr.width >= 64u && curr.height >= 64u) {
        // this call's visiting order (motion_order.hip: the hint kernel, which also clears the call's control area -- tile flags,
        // segment marks and map, counters, queue -- and the order kernel)
        uint32_t *hints = reinterpret_cast<uint32_t *>(workspace + l.order);
        uint32_t *callOrder = hints + kHints;
        e = launch_motion_order(s, prev, curr, hints, callOrder, entryOfScan, baseScan, flags, (int)((l.order - l.tileFlags) / sizeof(uint32_t)), framesInFlight);
        if (e != hipSuccess) return e;
        order = callOrder;
        verdictWord = reinterpret_cast<uint32_t *>(workspace + l.verdict);
        // the strips the frame's motion exposes, decided completely by a kernel of their own (motion_strip.hip), which lists what it took
        if (knobs.strips && !fused.data && strip_frames_ok(prev, curr, mv)) {
            uint32_t *colBand = reinterpret_cast<uint32_t *>(workspace + l.colBand), *rowBand = reinterpret_cast<uint32_t *>(workspace + l.rowBand);
            e = launch_motion_strip(s, prev, curr, mv, order, rank2scan, rankIsScan, colBand, rowBand, flags, sp.queueCount + 1, tilesX,
                                    knobs.stripPad >= 0 ? knobs.stripPad : (framesInFlight ? 0 : 20 * 1024));
            if (e != hipSuccess) return e;
            sp.colBand = colBand; sp.rowBand = rowBand;
        }
        // the whole interior tiles, through the lean kernel first (motion_lean.hip): what it settles it marks in segDone -- cleared by
        // the hint kernel above -- and the generic kernel below skips
        if (lean) {
            // (beside the persistent kernel on a stream of its own, with that kernel's grid cut to 448 .. 320 workgroups to leave it room:
            //  measured for a context that runs one frame at a time, 2,160 - 2,430 frames/s against 2,610 without the kernel: not done)
            e = launch_motion_lean(s, prev, curr, mv, order, rank2scan, rankIsScan, reinterpret_cast<const uint32_t *>(workspace + l.leanTiles), l.leanLaunch, sp.tilesX, segDone,
                                   reinterpret_cast<uint32_t *>(workspace + l.hardTiles), ctrl + kCtrlHardCount, ctrl + kCtrlLeanSettled, knobs.leanForce == 1);
            if (e != hipSuccess) return e;
        }
    } else {
        e = hipMemsetAsync(flags, 0, l.order - l.tileFlags, s);         // (no hints: the fixed order; the area is cleared by a memset)
        if (e != hipSuccess) return e;
    }
    int groups = std::max(1, std::min(sp.units, l.slots > 0 ? l.slots : sp.units));
    if (groupsCap > 0) groups = std::max(1, std::min(groups, groupsCap));      // (frames in flight: lfg_capi.cpp, motion_run)
    if (knobs.prefGroups > 0) groups = std::max(1, std::min(l.slots > 0 ? l.slots : groups, knobs.prefGroups));      // (measurement)
    auto launchPersistent = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(groups), dim3(kPNT), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int)curr.width, (int)curr.height, list, umin, count, flags, tilesX, order, sp,
                           (int8_t *)mv.data, (int)mv.pitch, rank2scan, segDone, ctrl);
    };
    if (fused.data) launchPersistent(motion_prefilter_kernel<true>);
    else if (tier == 1) launchPersistent(motion_prefilter_kernel<false, 1>);
    else launchPersistent(motion_prefilter_kernel<false>);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    LFG_STAMP(motion_stamps_report(s, sp);)
    const int segments = sp.tilesX * (((int)curr.height + kPTH - 1) / kPTH) * (kPTH / kSeg);
    // (a call that went through the lean kernel -- a pan, stills, moving objects -- takes the small grid with frames in flight too:
    //  pan +2.1 %, stills +3 %, occluded and moving objects +0.8 %; noise, which does not, loses 1.8 % by it and keeps the large one)
    int resolveGroups = (framesInFlight && !lean) ? segments * (kSeg / 4) : std::min(kResolveGroups, segments * (kSeg / 4));
    if (knobs.resolveGroups > 0) resolveGroups = std::max(1, std::min(segments * (kSeg / 4), knobs.resolveGroups));      // (measurement)
    e = launch_motion_resolve(s, prev, curr, mv, list, umin, count, flags, tilesX, sp, rank2scan, segDone, resolveGroups);
    if (e != hipSuccess) return e;
    e = launch_motion_tiled_8_16(s, prev, curr, mv, flags, rank2scan, reinterpret_cast<unsigned long long *>(workspace + l.merge),
                                 sp.queueCount + 1, fused, expectNoFallback && verdictWord != nullptr, verdictWord,
                                 verdictWord ? leanFlagHost : nullptr);
    // (the order kernel's verdict on this call's content (order32[kCand + 2]) reaches the host -- which decides with it how the lane's
    //  NEXT call is launched -- by a store of that last launch into the host's pinned word; until round 4's end a copy command
    //  behind the call, a dispatch of its own on the lane's stream: without it the pan is where it was, stills, moving objects and
    //  occlusions +0.5 %)
    return e;
}

}  // namespace lfg
