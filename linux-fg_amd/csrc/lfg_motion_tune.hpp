// Tuning constants of the prefiltered motion path (csrc/motion_prefilter.hip and its prefilter_*.inc pieces, motion_order.hip,
// motion_plan.hip, motion_resolve.hip) -- shaders/motion.comp:27-52.  Every value here was settled by measurement on the MI355X
// (NOTES_r01_r03.md, NOTES_r04.md carry the A/B figures); none of them changes a result, only the run time.
//
// These were `#ifndef LFG_X / #define LFG_X v / #endif` macros scattered over motion.hip until round 5 (101 preprocessor
// conditionals).  They are plain constants now and keep their names.  An experiment builds a variant from a COPY of this header:
//     tools/build_variant.sh NAME LFG_HEAD=3 LFG_EIGHT_MAX=600.0f
// rewrites the named constants in the copy and compiles the motion files against it (a -DLFG_HEAD=3 on the command line no
// longer compiles: the name would be replaced inside its own definition).
#pragma once

namespace lfg {

// ---- lists of recorded candidates (motion_prefilter.hip, "Recorded candidates per pixel and list")
constexpr int LFG_LIST_MAIN = 10;                   // whole tiles
constexpr int LFG_LIST_AUX = 10;                    // units that share a tile
constexpr int LFG_LIST_DYN = 10;                    // parts of a segment handed over at run time

// ---- the call's visiting order (motion_order.hip)
constexpr int LFG_HINT_GRID = 16;                   // sample blocks per axis: 256 hints
constexpr int LFG_HINT_THREADS = 256;               // threads of a hint workgroup beside other frames' launches
constexpr int LFG_HINT_THREADS_ALONE = 1024;        // a context that runs one frame at a time: nothing else wants the CU, and a call's latency is what counts
constexpr int LFG_HINTS_MAX = 2 + 62;               // entries of the order taken as hints at most

// ---- the persistent kernel and its work units
constexpr int LFG_PREF_OCC = 2;                     // waves per SIMD: 2 = 256 VGPRs, no spills (168 VGPRs at 3 spill 89 registers to scratch:
                                                    // +24 % on matched content, 6 % faster only where every segment searches in full)
constexpr int LFG_DYN_PARTS = 8;                    // parts of the candidate order a handed-over segment is searched in, AT MOST (4 or 8): what the scratch is
                                                    // laid out for.  A call uses sp.dynParts: eight when one frame runs at a time (two workgroups halve the
                                                    // longest units of a launch: occluded 841 -> 890 frames/s, moving objects 983 -> 1,074), four with
                                                    // frames in flight, where the sum of all units' times is what counts and a part's staging and setup
                                                    // are paid half as often (occluded 1,047 -> 1,082, moving objects 1,272 -> 1,281).
constexpr bool LFG_QUEUE_INIT = true;               // the parts of a handed-over segment start from the thresholds of the wave that handed it over
constexpr bool LFG_QUEUE_FIRST = false;             // queued segments before plan units
constexpr int LFG_PREF_POLL_SLEEPS = 2;             // x 8128 clocks between two looks of a waiting workgroup
constexpr bool LFG_BORDER_PER_SEGMENT = true;       // "at the image border" decided per segment, not per tile
constexpr int LFG_HEAD = 2;                         // entries of the order a unit that shares its tile runs first, for its thresholds (prefilter_unit)
constexpr int LFG_FIRST_BATCH = 1;                  // entries of the first batch: the top hint alone (2: with zero motion, as in round 1)
constexpr bool LFG_LATE_THRESHOLDS = true;          // a segment unit that pools its four parts writes thresholds and counts only if a pixel stays open
constexpr bool LFG_LEAN_XCD_BANDS = true;           // workgroup i of the lean launch lands on XCD i mod 8: give every XCD a contiguous band of tiles, in raster
                                                    // order, so that the windows of neighbouring tiles (they overlap 2.7 x) meet in ONE L2
constexpr bool LFG_LEAN_PARTIAL = true;             // the lean kernel takes the rim tiles' inner segments as well

// ---- the batch loop (prefilter_batches.inc)
constexpr bool LFG_RANK_ALWAYS = true;              // a wave with the whole order turns to the ranks as soon as it can
constexpr bool LFG_RANK_ARITH = true;               // by rank, the window offset of a candidate from its rank by arithmetic (see candidateAt)
constexpr int LFG_LOOKAHEAD = 3;                    // candidates per lane of the lookahead (0 or 1: off; 2 until the ranks gave the window offsets by arithmetic)
constexpr bool LFG_EXACT_MATCH = true;              // a hint whose every block position is the same bytes in both frames skips its evaluation
constexpr int LFG_DEFER_FROM = 0;                   // a batch's survivors wait in the list when they are more than this many (0: always -- a
                                                    // lone survivor evaluated on the spot is an evaluation without anything to overlap with: window
                                                    // reads, column sums, slab round trip and row sums one after the other; a dozen together at
                                                    // the end of the search run software-pipelined.  8 until round 3: noisy frames +3 % with
                                                    // frames in flight, +8 % one at a time)
constexpr int LFG_SIXTEEN_FROM = 12;                // ... and the sixteen-point test runs on more than this many of them (its 4,400 instructions
                                                    // are a dozen evaluations)

// ---- the lattice tests (prefilter_tests.inc, prefilter_walks.inc): which applies below which largest threshold of the wave
constexpr bool LFG_ZERO_COMPARE = true;             // exact-texel compares once every threshold stands for a zero cost
constexpr bool LFG_SAD_TEST = true;                 // one-point lattice test by SAD while the wave's largest threshold is small
constexpr float LFG_SAD_TEST_MAX = 8.0f;
constexpr float LFG_ONEPOINT_MAX = 32.0f;           // below: the one-point test alone
constexpr float LFG_ONEPOINT_OFF = 96.0f;           // above: no one-point test (a distance is at most 510, but few exceed a threshold of a hundred)
constexpr float LFG_FOURPOINT_MAX = 4.0f * 510.0f;
constexpr int LFG_ORDER_BY_BANK = 1;                // the fixed visiting order dealt out by LDS bank (motion_order.hip: motion_tables)
constexpr float LFG_FOUR_SAD_MAX = 300.0f;          // the four-point walk by SADs below this
constexpr float LFG_EIGHT_SAD_MIN = 220.0f;         // (below it the four-point walk by SADs as in the default kernel; 300: noise of +-3 levels 1,180 instead of 1,390 frames/s, 150: +-2 1,420 instead of 1,590)
constexpr float LFG_EIGHT_SAD_MAX = 768.0f;         // kTier 1 (motion_prefilter_kernel<false, 1>): from there up to this the EIGHT-point walk by SADs takes the four-point walk's place
constexpr int LFG_EIGHT_SAD_OFF_AT = 48;            // ... until a full batch comes out of it with this many candidates left
constexpr int LFG_FOUR_OFF_AT = 48;                 // survivors of a full batch at which a wave gives the four-point walk up
constexpr int LFG_FOUR_MIN_WIDE = 2;                // candidates of a batch from which the walk pays where the one-point test does not apply
constexpr bool LFG_SIXTEEN = true;
constexpr float LFG_SIXTEEN_MAX = 2048.0f;
constexpr bool LFG_EIGHT = true;
constexpr float LFG_EIGHT_MAX = 512.0f;             // (1,024: the top and right rim of the pan, thresholds of 620 - 700, let more than a dozen of a batch
                                                    //  through and pay for both walks -- pan -1.3 %; 512: pan +1.3 %, noisy +2 %, occluded +1.1 %,
                                                    //  moving objects +0.8 %; 400 and 600 within 0.5 % of it)
constexpr float LFG_EIGHT_MAX_INSIDE = 1024.0f;     // ... away from the border, where such thresholds are heavy noise everywhere (+-8 levels at the input:
                                                    //  575 -> 682 frames/s) and not a rim's few rows; a wave whose first full batch leaves more than the
                                                    //  sixteen-point walk's worth (LFG_EIGHT_OFF_AT) stops trying
constexpr int LFG_EIGHT_OFF_AT = 16;
constexpr bool LFG_BAND = true;                     // the walks cover only the columns that hold unsettled pixels ("THE BAND")

// ---- narrow search and row band (prefilter_narrow.inc, prefilter_rowband.inc)
constexpr bool LFG_NARROW = true;
constexpr bool LFG_ROW_BAND = true;
constexpr float LFG_NARROW_THR = 2048.0f;           // a pixel whose threshold is still this large after the hints has no match
                                                    // (below it the sixteen-point test still drops wrong candidates)

}  // namespace lfg
