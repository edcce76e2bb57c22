// Geometry and constants of the prefiltered motion path shared by csrc/motion_prefilter.hip (the generic persistent kernel and everything
// around it) and csrc/motion_lean.hip (the lean kernel for whole interior tiles).  blockSize 8, searchRadius 16:
// shaders/motion.comp:16-57 as dispatched by src/frame_manager.cpp:325-344.
#pragma once

namespace lfg {

constexpr int kB = 8, kR = 16;
constexpr int kSide = 2 * kR + 1;
constexpr int kCand = kSide * kSide;

constexpr int kPTW = 56, kPTH = 64;               // prefilter tile (pixels): 56 + 7 = 63 position columns <= 64 lanes
constexpr int kPNT = 256;
constexpr int kSeg = 16;                          // pixel rows per wave
constexpr int kSegD = kSeg + kB - 1;              // 23 distances per thread and candidate
constexpr int kWinW = 95;                         // window columns: 63 positions + 2R
constexpr int kWinH = kPTH + kB - 1 + 2 * kR;     // 103 rows; the window is stored COLUMN-major (pitch kWinH, odd), so
                                                  // a thread's 23 texels are consecutive words (ds_read2_b32 pairs) and
                                                  // the 64 lanes of a read still fall into distinct banks
// The one-point lattice of interior segments: block positions (kLatC0 + 8 i, kLatR0 + 8 j).  A pixel's block is the 8 x 8
// positions that start at its own column and row, so rows 7 | 15 put exactly one lattice row into the block of each of a
// segment's 16 pixel rows, and columns 7, 15 .. 55 one lattice column into the block of each of its 56 pixel columns:
// 2 x 7 = 14 points.  (Rows 3 | 11 | 19 and columns 3 .. 59 did the same with 3 x 8 = 24 until late in round 2.)
constexpr int kLatC0 = 7, kLatCols = 7, kLatR0 = 7, kLatRows = 2;
static_assert(kLatC0 + 8 * (kLatCols - 1) == kPTW - 1 && kLatR0 + 8 * (kLatRows - 1) == kSeg - 1 && kLatC0 == kB - 1 && kLatR0 == kB - 1,
              "one lattice point in every pixel's block: the first at the far end of pixel 0's, the last at the near end of the last pixel's");
constexpr int kRun = 7;                           // pixels per row-sum run: 8 runs x 7 = 56
constexpr int kRunIn = kRun + kB - 1;             // 13
constexpr float kRatio = 1.00008f;      // >= (1 + 3.6e-5) / (1 - 3.6e-5) with room for the product's rounding ("Bracket", motion_prefilter.hip)
constexpr float kRestart = 0.9997f;     // S~ < thr kRestart: every earlier record of the pixel is dead ((1 - 2^-13) / kRatio^2 = 1 - 2.8e-4, with room)
static_assert(kRestart < (1.0 - 1.0 / 8192.0) / (1.00008 * 1.00008) - 1e-5, "restart rule");
static_assert(kPTW + kB - 1 <= 64 && kWinW >= kPTW + kB - 1 + 2 * kR, "one lane per position column");
static_assert(kCand < 2048, "a rank fits eleven bits");
// The one-point test alone decides while the wave's largest threshold is below this (lfg_motion_tune.hpp: LFG_ONEPOINT_MAX); the lean
// kernel keeps a segment only while that holds.
constexpr float kOnePointOnly = 32.0f;
// The strip kernel (motion_strip.hip) decides the outermost kStripCols pixel columns (left, right) and kStripRows pixel rows (top,
// bottom) that a frame's motion exposes -- a translation by t exposes |t| pixels, and the blocks of three more reach into them: twelve
// columns cover |tx| <= 9 at the output resolution, eight rows |ty| <= 5 (what lies beyond stays with the persistent kernel) -- and says
// so per pixel row / column: colBand[y] bit 0 = the left band of row y, bit 1 = the right band; rowBand[x] bit 0 = the top band of
// column x, bit 1 = the bottom band (lfg_motion_common.hpp: strip_decided).
constexpr int kStripCols = 12, kStripRows = 8;
constexpr float kSadTestMax = 8.0f;     // ... by sums of absolute differences below this (lfg_motion_tune.hpp: LFG_SAD_TEST_MAX)

}  // namespace lfg
