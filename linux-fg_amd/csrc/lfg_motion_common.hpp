// What the translation units of the motion stage share (shaders/motion.comp:16-57): the arithmetic of one distance, the
// geometry of the literal kernel's tiles (the prefilter flags tiles on that grid), hints and list depths, the four-byte
// record of a candidate, and the small device helpers.  Geometry of the prefilter's own tiles: lfg_motion_tile.hpp;
// tuning constants: lfg_motion_tune.hpp; diagnostic stamps (empty in the shipped build): lfg_motion_diag.hpp.
//   motion_literal.hip    the shader's own 64-term chain for every (pixel, candidate): whole frames, flagged tiles, any B / R
//   motion_order.hip      this call's visiting order: hint kernel, order kernel, the candidate tables
//   motion_lean.hip       whole interior tiles whose pixels all find their answer in the call's first hints
//   motion_prefilter.hip  the persistent kernel: every other work unit (prefilter_*.inc: one unit, phase by phase)
//   motion_resolve.hip    the literal chain for the candidates the prefilter could not separate
//   motion_plan.hip       host side: work-unit tables and the workspace layout
#pragma once

#include "lfg_device.hpp"
#include "lfg_internal.hpp"
#include "lfg_interp.hpp"
#include "lfg_motion_tile.hpp"
#include "lfg_motion_tune.hpp"
#include "lfg_motion_diag.hpp"

namespace lfg {


// Correctly rounded sqrtf for the inputs this kernel produces: x = 0, or a sum of four squares in
// [(1/255)^2, 4].  One Newton step on v_rsq_f32 with an FMA residual (Markstein's form):
//     y = rsq(x),  g = x*y,  h = y/2,  r = x - g*g (exact in the FMA),  result = g + r*h.
// Unlike the compiler's IEEE sqrtf (v_sqrt_f32 + two residual tests + denormal scaling, ~23 VALU-op
// equivalents, tools/microbench.hip) this is 5 plain ops and one transcendental.  It is not proven
// correctly rounded in general; it IS verified exhaustively: lfg_selftest_sqrt compares it on the
// device with __builtin_sqrtf for every float in [2^-21, 8] and for 0
// (tests/test_gpu_parity.py::test_exact_sqrt_exhaustive: 0 mismatches in 201,326,593 values).
__device__ __forceinline__ float exact_sqrt(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y;
    const float h = 0.5f * y;
    const float r = __builtin_fmaf(-g, g, x);
    const float g2 = __builtin_fmaf(r, h, g);
    return x == 0.0f ? 0.0f : g2;          // rsq(0) = inf
}

// (The kernels that decide vectors are templates on kFused: the mere presence of this code in the default kernels -- a larger
//  plan structure, a few scalar tests in the epilogues -- cost the default path 2-3 %, measured; instantiated twice, it costs nothing.)
// The north-star order (lfg_internal.hpp: FusedOut): one pixel of the generated frame from the vector just decided for it --
// interpolate_pixel is the function the interpolate kernel is made of (csrc/lfg_interp.hpp), its uv the same fp32 division
// the kernel's tables hold, so the bytes are those of lfg_motion followed by lfg_interpolate.
__device__ __forceinline__ void fused_pixel(const FusedOut &fo, const uint8_t *__restrict__ prev, int prevPitch,
                                            const uint8_t *__restrict__ curr, int currPitch, int W, int H, int px, int py, int dx, int dy) {
    float mx = (float)dx, my = (float)dy;
    if (fo.intended) { mx = mx / (float)W; my = my / (float)H; }
    *reinterpret_cast<uint32_t *>(fo.data + (size_t)py * (size_t)fo.pitch + (size_t)px * 4u) =
        interpolate_pixel(prev, prevPitch, curr, currPitch, W, H, px, py, mx, my, fo.t);
}

// distance() of two texels already converted to float, oracle choice (7):
// sqrt(((dx*dx + dy*dy) + dz*dz) + dw*dw), correctly rounded sqrt.
template <bool FAST_SQRT>
__device__ __forceinline__ float dist4(const float (&c)[4], f32x4 p) {
    const float dx = c[0] - p.x, dy = c[1] - p.y, dz = c[2] - p.z, dw = c[3] - p.w;
    const float s = ((dx * dx + dy * dy) + dz * dz) + dw * dw;
    return FAST_SQRT ? exact_sqrt(s) : __builtin_sqrtf(s);
}


// Has the strip kernel decided pixel (px, py)?  cb = colBand[py], rb = rowBand[px] (lfg_motion_tile.hpp).
__device__ __forceinline__ bool strip_decided(uint32_t cb, uint32_t rb, int px, int py, int W, int H) {
    return ((cb & 1u) != 0u && px < kStripCols) || ((cb & 2u) != 0u && px >= W - kStripCols) ||
           ((rb & 1u) != 0u && py < kStripRows) || ((rb & 2u) != 0u && py >= H - kStripRows);
}

// The literal kernel's tiles (motion_literal.hip): 64 x 64 pixels, 512 threads.  The prefilter flags tiles on THIS grid.
constexpr int kTW = 64, kTH = 64;                 // pixel tile
constexpr int kNT = 512;                          // threads per workgroup: one per 8x1 pixel patch
constexpr int kShareBelow = 256;        // flagged tiles up to which the exact kernel shares each between several workgroups
constexpr int kFallbackParts = 8;       // workgroups that share a flagged tile (contiguous parts of the tie order)
constexpr int kResolveGroups = 1024;    // workgroups of the resolve kernel's small grid (motion_resolve.hip)

constexpr int kHintGrid = LFG_HINT_GRID, kHints = kHintGrid * kHintGrid;      // sample blocks of the per-call visiting order
// Recorded candidates per pixel and list.  A pseudo-random visiting order makes the number of running minima of a pixel
// without any match ~Poisson(ln n) over the n candidates a wave visits, whatever the content -- 7.6 for a whole tile's wave,
// and measured so: on the hand-over test's 4K frame (every segment handed over, 7.3 million pixel-parts) the eight parts hold
// 2.8 records on average where sum 1 / (8 + j) says 2.84, P(>= 10) = 7e-4 for 5e-4, P(>= 14) = 1.5e-6 for 8e-7 -- and two lists
// of 17 and 18, genuine runs of successive minima (tools/debug_dyn_lists.py prints them), a hundred times what the model's
// far tail allows: what gave up at depth 16 in round 2, when the lists had to hold every record of a search (32 / 24 / 24:
// 2.4 GB of workspace at 4K).  They do not have to.  A record whose cost bound exceeds the pixel's CURRENT threshold can
// never pass the resolve kernel's test against the final, smaller one; and when a candidate undercuts the threshold by
// more than the bracket is wide -- S~ < thr (1 - 3e-4): every earlier record has S~ >= the old minimum = thr / kRatio, its
// stored bound is at most 2^-13 below that, and kRatio S~ lies under it -- ALL earlier records are dead at once, so the
// pixel's count restarts at 0 (the record path, `restart`): no read, no pass over the list.  On content without ties
// successive minima differ by a percent, not by 0.03 %: a list holds the running minimum and the odd near-tie, whatever
// the length of the search (records held per pixel at the end of a noise frame: 1.3; recorded over it: 7.5).  What fills
// a list now is a set of candidates within 0.03 % of each other -- ties: flat or periodic content -- and that flags the
// tile for the literal kernel as before.  Depth 10 everywhere (one slot of it a spare: see listsOverflowed).
constexpr int kListK = LFG_LIST_MAIN, kListAux = LFG_LIST_AUX, kListDyn = LFG_LIST_DYN;
static_assert(kListK >= 4 && kListAux >= 4 && kListDyn >= 4 && kListK <= 64 && kListAux <= 64 && kListDyn <= 64, "list depths");
static_assert((kHints & (kHints - 1)) == 0 && kHints >= 256 && kHints <= 1024, "one hint per thread of the order kernel, scrambled by an odd multiplier");
static_assert(kPNT / 64 * kSeg == kPTH && 8 * kRun == kPTW && kPTH == kTH, "stage maps cover the tile");


// Candidates whose whole shifted block lies outside prev on one axis sample nothing but out-of-image zeros, so for
// a given pixel they all cost exactly the same (the same sequence of |curr texel| distances): a plateau of up to a few
// hundred tied candidates next to the left/right/top/bottom edge.  Only the first of them in tie order can win, so the
// prefilter records ONE member of the plateau per pixel and the resolve kernel replaces it by the first in tie order.
// ... for some pixel of the rectangle [x0, x1] x [y0, y1]
__device__ __forceinline__ bool block_leaves_prev_any(int x0, int x1, int y0, int y1, int dx, int dy, int W, int H) {
    return (x0 + kB / 2 - 1 + dx < 0) | (x1 - kB / 2 + dx >= W) | (y0 + kB / 2 - 1 + dy < 0) | (y1 - kB / 2 + dy >= H);
}
__device__ __forceinline__ bool block_leaves_prev(int px, int py, int dx, int dy, int W, int H) {
    return (px + kB / 2 - 1 + dx < 0) | (px - kB / 2 + dx >= W) | (py + kB / 2 - 1 + dy < 0) | (py - kB / 2 + dy >= H);
}

// A recorded candidate: 21 bits of its S~ (the float's exponent and 13 mantissa bits, i.e. S~ rounded DOWN by at most
// 2^-13 of its value) above its 11-bit rank in the tie order.  The resolve kernel keeps a record while that lower
// bound passes the pixel's final threshold: no survivor is lost, and a record within 0.012 % above the threshold is
// kept needlessly and goes through the literal chain with the others.  Four bytes instead of eight: half the scratch
// and half the bytes a record moves.
typedef uint32_t Rec;
__device__ __forceinline__ Rec rec_make(float s, uint32_t cand) { return ((__builtin_bit_cast(uint32_t, s) >> 10) << 11) | cand; }
__device__ __forceinline__ float rec_cost_low(Rec r) { return __builtin_bit_cast(float, (r >> 11) << 10); }
__device__ __forceinline__ uint32_t rec_cand(Rec r) { return r & 0x7FFu; }

typedef const __attribute__((address_space(3))) uint32_t *lds_ro_u32_ptr;
typedef const __attribute__((address_space(3))) float *lds_ro_f32_ptr;


constexpr uint32_t kNoUnit = 0xFFFFFFFFu;
// A word that workgroups on other XCDs update with atomics, as they left it: an agent-scope load (served by L2, which
// those atomics write through), and every 32nd look a read-modify-write that changes nothing -- a compare-and-swap
// against a value the word never holds -- so that progress never depends on a cache line being refreshed.
__device__ __forceinline__ uint32_t peek(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t peek_hard(uint32_t *p) { return atomicCAS(p, 0xFFFFFFFFu, 0xFFFFFFFFu); }


}  // namespace lfg
