// motion.hip -- per-pixel full-search block matching, the MI355X-native replacement of
// shaders/motion.comp (reference: /root/reference/shaders/motion.comp:16-57, dispatched by
// src/frame_manager.cpp:325-344 with blockSize = 8, searchRadius = 16.0f).
//
// For every pixel p and every candidate m in [-R,R]^2 (dy outer, dx inner, scan starts at (-R,-R)):
//     cost(p,m) = sum over the BxB block c = p - B/2 + (x,y), row-major, c inside the image, of
//                 distance(curr(c), prev(c+m)),        prev out of bounds -> (0,0,0,0)
// and the FIRST candidate reaching the minimum wins (strict '<', motion.comp:49).  The result must be
// bit-exact, and the fp32 sum is a single sequential chain, so no re-association is allowed: the
// tiled kernel keeps the literal summation order and only shares the per-pixel distances
//     d_m(c) = distance(curr(c), prev(c+m))
// between the up to B*B pixels whose blocks contain c.  Skipped (out-of-image) positions are added
// as +0.0f, which leaves a non-negative fp32 partial sum unchanged, so the chain is uniform.
//
// Roofline: fp32 VALU, not HBM: 64 dependent-order adds per (pixel, candidate) plus one
// correctly rounded distance per (position, candidate); HBM traffic is 10 B/pixel.
//
// motion_tiled_8_16_kernel (B = 8, R = 16): a 512-thread workgroup owns a 64x64 pixel tile.
//   Texels come in through TYPED buffer loads (tbuffer_load_format_xyzw, RGBA8 UNORM): the
//   texture-address unit returns four floats that are bit-exact byte/255.0f (tools/probe_unorm.hip),
//   so UNORM conversion costs no VALU work, and rows outside the image load zeros by the buffer's own
//   range check.  prev is read straight from L1/L2 (the tile's search window is ~42 KB, re-read by
//   every candidate); LDS holds only a double-buffered plane D[71][72] of distances for ONE candidate
//   over the tile plus its block halo (41 KB: two workgroups per CU, 4 waves/SIMD).
//   Per candidate k:   phase A  every thread computes 10 entries of D_{k+1}: one column of nine rows plus
//                               one entry of the seven leftover columns (its curr texels stay in
//                               registers as floats for the whole kernel), ten typed loads in one batch
//                      phase B  every thread owns 8 horizontally adjacent pixels of one row and runs
//                               their 8 chains over D_k: 8 rows x 15 floats from LDS, 512 adds
//                      one __syncthreads().
// motion_generic_kernel: any block size / whole-number radius, one thread per pixel, literal loops
//   straight from global memory (slow; used for non-default parameters and as an on-device cross-check).
#include "lfg_device.hpp"
#include "lfg_internal.hpp"
#include "lfg_interp.hpp"
#include "lfg_motion_tile.hpp"

#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

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

// ------------------------------------------------------------------------------ tiled, B = 8, R = 16

constexpr int kTW = 64, kTH = 64;                 // pixel tile
constexpr int kNT = 512;                          // threads per workgroup: one per 8x1 pixel patch
constexpr int kDW = kTW + kB - 1;                 // 71 block positions across
constexpr int kDH = kTH + kB - 1;                 // 71 down
constexpr int kDS = 74;                           // D row pitch in LDS (floats): 74/2 = 37 = 1 mod 4 makes the
                                                  // chain phase's ds_read_b64 pattern conflict-free (see phase B)
// Phase-A ownership of the 71 x 71 distance plane: thread (lane, wave g) computes column `lane` of rows
// 9g .. 9g+8 (the main 64 x 72 block; row 71 does not exist), and threads 0..496 one entry each of the
// remaining 7 columns.  Ten entries per thread, but only two base addresses to keep: the nine main
// entries are one column, so their prev offsets differ by whole (wave-uniform) row pitches and their LDS
// addresses by a constant.
constexpr int kMainRows = 9;
constexpr int kExtraCols = kDW - 64;              // 7
constexpr int kExtra = kExtraCols * kDH;          // 497 entries
constexpr int kPos = kMainRows + 1;               // 10
static_assert(kTW / 8 * kTH == kNT, "one thread per 8x1 pixel patch");
static_assert(kNT / 64 * kMainRows >= kDH && kExtra <= kNT, "phase-A map covers the plane");

constexpr int kShareBelow = 256;        // flagged tiles up to which the exact kernel shares each between several workgroups
constexpr int kFallbackParts = 8;       // workgroups that share a flagged tile (contiguous parts of the tie order)
// One tile (or one part of a flagged tile) of the literal kernel: see motion_tiled_8_16_kernel, which decides what this workgroup takes.
template <bool kFused>
__device__ __forceinline__ void exact_tile(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H, const uint32_t *__restrict__ rank2scan, unsigned long long *__restrict__ merge,
    uint32_t *__restrict__ flaggedTiles, const FusedOut &fo, const int tileX, const int tileY, const int parts, const int part, const int slot,
    float (&sD)[2][kDH * kDS], uint32_t &sLast) {
    constexpr int kOob = (int)0x80000000;        // a buffer offset that always fails the range check
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);           // wave index 0..7
    const int tx0 = tileX * kTW, ty0 = tileY * kTH;                   // tile origin (pixels)
    const int bx0 = tx0 - kB / 2, by0 = ty0 - kB / 2;                 // image coords of D(0,0)
    const i32x4 rPrev = make_rgba8_rsrc(prev, (uint32_t)H * (uint32_t)prevPitch);
    const i32x4 rCurr = make_rgba8_rsrc(curr, (uint32_t)H * (uint32_t)currPitch);
    // Interior tiles: every block position lies inside the image and no candidate can leave it sideways.
    // Rows above/below the image always fall out of the buffer range and load zeros by themselves.
    const bool interior = __builtin_amdgcn_readfirstlane(
        (bx0 - kR >= 0) && (bx0 + kDW - 1 + kR < W) && (by0 >= 0) && (by0 + kDH - 1 < H));

    // ---- this thread's ten block positions and their curr texels (kept as floats for the whole kernel)
    const int gxM = bx0 + lane, gyM0 = by0 + kMainRows * g;           // main block: column, first row
    const bool colOkM = gxM >= 0 && gxM < W;
    const bool hasE = tid < kExtra;
    const int cyE = tid / kExtraCols, cxE = 64 + tid - cyE * kExtraCols;
    const int gxE = bx0 + cxE, gyE = by0 + cyE;
    const bool okE = hasE && gxE >= 0 && gxE < W && gyE >= 0 && gyE < H;
    const int pb0 = (gyM0 - kR) * prevPitch + (gxM - kR) * 4;         // prev(c + (-R,-R)), main row 0 (may be < 0)
    const int pbE = (gyE - kR) * prevPitch + (gxE - kR) * 4;
    float *const dM = &sD[0][kMainRows * g * kDS + lane];             // LDS slot of main row 0, buffer 0
    float *const dE = &sD[0][cyE * kDS + cxE];
    float cf[kPos][4];
    {
        int co[kPos];
#pragma unroll
        for (int j = 0; j < kMainRows; ++j) {
            const int gy = gyM0 + j;
            co[j] = (colOkM && gy >= 0 && gy < H && kMainRows * g + j < kDH) ? gy * currPitch + gxM * 4 : kOob;
        }
        co[kMainRows] = okE ? gyE * currPitch + gxE * 4 : kOob;
        f32x4 c4[kPos];
        load_rgba8_unorm_x10(c4, co, rCurr);
#pragma unroll
        for (int n = 0; n < kPos; ++n) { cf[n][0] = c4[n].x; cf[n][1] = c4[n].y; cf[n][2] = c4[n].z; cf[n][3] = c4[n].w; }
    }

    // ---- phase A: D(c) = distance(curr(c), prev(c + m)) for one candidate, 0 for c outside the image.
    auto phaseA = [&](int cand, int buf) {
        const int scan = (int)rank2scan[cand];
        const int dyi = scan / kSide, dxi = scan - dyi * kSide;       // dy + R, dx + R (wave-uniform)
        const int candOff = dyi * prevPitch + dxi * 4;
        int o[kPos];
        if (interior) {
            const int b = pb0 + candOff;
#pragma unroll
            for (int j = 0; j < kMainRows; ++j) o[j] = b + j * prevPitch;
            o[kMainRows] = hasE ? pbE + candOff : kOob;
        } else {
            // prev(c + m) left/right of the image -> zero texel (its row offset would alias a neighbour row)
            const int b = (unsigned)(gxM - kR + dxi) < (unsigned)W ? pb0 + candOff : kOob;
#pragma unroll
            for (int j = 0; j < kMainRows; ++j) o[j] = b + j * prevPitch;   // kOob + j*pitch stays out of range
            o[kMainRows] = (hasE && (unsigned)(gxE - kR + dxi) < (unsigned)W) ? pbE + candOff : kOob;
        }
        f32x4 p[kPos];
        load_rgba8_unorm_x10(p, o, rPrev);
        const int bo = buf * (kDH * kDS);
#pragma unroll
        for (int j = 0; j < kMainRows; ++j) {
            float d = dist4<true>(cf[j], p[j]);
            if (!interior) {
                const int gy = gyM0 + j;
                d = (colOkM && gy >= 0 && gy < H) ? d : 0.0f;         // block position outside the image: skipped
            }
            if (kMainRows * g + j < kDH) dM[bo + j * kDS] = d;        // wave-uniform: row 71 does not exist
        }
        if (hasE) {
            const float d = dist4<true>(cf[kMainRows], p[kMainRows]);
            dE[bo] = (interior || okE) ? d : 0.0f;
        }
    };

    // Phase-B ownership: 8 pixels (tile row `ry`, columns 8*rxq .. 8*rxq+7).
    const int rxq = tid & 7, ry = tid >> 3;
    float best[8];
    int bestCand[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { best[i] = 1e10f; bestCand[i] = 0; }   // motion.comp:23-24

    // (many flagged tiles fill the chip by themselves: then each is one workgroup's again -- sharing costs a fifth more in
    //  repeated set-up -- and the vectors are written directly)
    const int perPart = (kCand + parts - 1) / parts;
    const int candBegin = part * perPart, candEnd = min(kCand, candBegin + perPart);
    phaseA(candBegin, candBegin & 1);
    __syncthreads();

    for (int cand = candBegin; cand < candEnd; ++cand) {
        phaseA(cand + 1, (cand + 1) & 1);   // cand + 1 == kCand reads past the last row: zeros, never used
        __builtin_amdgcn_sched_barrier(0);  // keep phase A's loaded texels out of phase B's live range

        // ---- phase B: the 8 sequential chains of this thread's pixels, block rows top to bottom,
        //      block columns left to right -- the literal order of motion.comp:33-47.
        // One block row per step: eight 8-byte LDS reads (15 of the 16 floats are used) and 64 adds, the
        // next row's reads issued before this row's adds.  With a 74-float pitch the 32 lanes of a
        // ds_read_b64 group (4 tile rows x 8 patches) hit 32 distinct 8-byte slots of the 256-byte bank
        // row -- (37*row + 4*patch + j) mod 32 is a bijection -- so the reads are conflict-free (16-byte
        // reads on a 72-float pitch were 3-way conflicted, 63 % of the LDS cycles, and no pitch fixes
        // ds_read_b128's lane groups).  The reads are volatile so the compiler keeps them as eight
        // ds_read_b64: merged into ds_read2_b64 they bank differently and conflict again.
        typedef const volatile __attribute__((address_space(3))) f32x2 *lds_f32x2_ptr;   // keep it a DS access
        const lds_f32x2_ptr rowp = (lds_f32x2_ptr)(sD[cand & 1] + ry * kDS + rxq * 8);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        f32x2 q[8], nq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = rowp[j];
#pragma unroll
        for (int y = 0; y < kB; ++y) {
            if (y + 1 < kB) {
#pragma unroll
                for (int j = 0; j < 8; ++j) nq[j] = rowp[(y + 1) * (kDS / 2) + j];
            }
            const float e[16] = {q[0].x, q[0].y, q[1].x, q[1].y, q[2].x, q[2].y, q[3].x, q[3].y,
                                 q[4].x, q[4].y, q[5].x, q[5].y, q[6].x, q[6].y, q[7].x, q[7].y};
#pragma unroll
            for (int x = 0; x < kB; ++x) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += e[i + x];       // pixel i, block column x
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = nq[j];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (acc[i] < best[i]) { best[i] = acc[i]; bestCand[i] = cand; }   // motion.comp:49-52
        }
        __syncthreads();
    }

    const int py = ty0 + ry, px0 = tx0 + rxq * 8;
    if (parts > 1) {
        // (row ry of every slot side by side: a tile's 64 rows of words lie 128 KB apart, over all memory channels -- the 4096
        //  atomics a workgroup ends with, eight workgroups per tile, queued up on a few channels when a slot's 32 KB were contiguous)
        unsigned long long *const words = merge + ((size_t)ry * (size_t)kShareBelow + (size_t)slot) * (size_t)kTW + (size_t)(rxq * 8);
        // RETURNING atomics, and their results feed the barrier: a device-scope atomic has been performed where every XCD
        // sees it by the time it returns, so "all of this workgroup's words are in" needs no fence -- __threadfence() writes
        // back and invalidates the XCD's whole L2, and with one in every part's tail the workgroups still searching kept
        // losing their windows (the launch 6 % longer, measured)
        unsigned long long seen = 0ull;
        if (py < H) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (px0 + i < W)
                    seen |= atomicMin(&words[i], ((unsigned long long)__builtin_bit_cast(uint32_t, best[i]) << 32) | (unsigned long long)(uint32_t)bestCand[i]);
            }
        }
        asm volatile("" : : "v"(seen));
#ifdef LFG_DIAG_NO_MERGE_TAIL          // (timing experiment, wrong results: the parts leave without merging)
        return;
#endif
        // the part that arrives last (a counter per slot) writes the tile's vectors and leaves the slot's words all ones for
        // the next call
        __syncthreads();
        if (tid == 0) sLast = atomicAdd(flaggedTiles + 1 + kShareBelow + slot, 1u) == (uint32_t)(parts - 1) ? 1u : 0u;
        __syncthreads();
        if (sLast == 0u) return;
        if (py < H) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (px0 + i < W) {
                    const unsigned long long word = __hip_atomic_load(&words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&words[i], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bestCand[i] = (int)(uint32_t)word;
                }
            }
        }
    }
    if (py < H && px0 < W) {
        uint32_t o[4] = {0u, 0u, 0u, 0u};                             // 8 x (int8 dx, int8 dy)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int scan = (int)rank2scan[bestCand[i]];
            const int dyi = scan / kSide, dxi = scan - dyi * kSide;
            const uint32_t v = (uint32_t)(uint8_t)(int8_t)(dxi - kR) | ((uint32_t)(uint8_t)(int8_t)(dyi - kR) << 8);
            o[i >> 1] |= v << (16 * (i & 1));
            if (kFused && fo.data && px0 + i < W) fused_pixel(fo, prev, prevPitch, curr, currPitch, W, H, px0 + i, py, dxi - kR, dyi - kR);
        }
        int8_t *dst = mv + (size_t)py * (size_t)mvPitch + (size_t)px0 * 2u;
        if (kFused && !fo.storeMv) {
        } else if (px0 + 7 < W && (mvPitch & 15) == 0) {
            *reinterpret_cast<uint4 *>(dst) = uint4{o[0], o[1], o[2], o[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (px0 + i < W) {
                    const uint32_t v = o[i >> 1] >> (16 * (i & 1));
                    dst[2 * i] = (int8_t)(v & 0xff); dst[2 * i + 1] = (int8_t)((v >> 8) & 0xff);
                }
            }
        }
    }
}

// What a workgroup of the second pass takes: item -> (tile, parts, part, slot); false: nothing.
__device__ __forceinline__ bool fallback_item(int item, uint32_t flagged, const uint32_t *__restrict__ tileFlags, const uint32_t *__restrict__ flaggedTiles,
                                              int tilesX, int tiles, int &tileX, int &tileY, int &parts, int &part, int &slot) {
    int t;
    parts = 1; part = 0; slot = 0;
    if (flagged <= (uint32_t)kShareBelow) {
        // (slot fastest: workgroups go round the XCDs in launch order, so the eight parts of a slot land on ONE XCD and
        //  share the tile's search window in its L2 -- part fastest put them on eight and cost a quarter more, measured;
        //  a grid that loops has to be a multiple of eight for that: it is)
        static_assert(kShareBelow % 8 == 0, "a slot's parts on one XCD");
        parts = kFallbackParts; slot = item % kShareBelow; part = item / kShareBelow;
        if ((uint32_t)slot >= flagged || part >= kFallbackParts) return false;
        t = (int)flaggedTiles[1 + slot];
    } else {
        t = item;
        if (t >= tiles || tileFlags[t] == 0u) return false;
    }
    tileY = t / tilesX; tileX = t - tileY * tilesX;
    return true;
}

template <bool kFused>
__global__ __launch_bounds__(kNT, 4) void motion_tiled_8_16_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H, const uint32_t *__restrict__ tileFlags,
    const uint32_t *__restrict__ rank2scan, unsigned long long *__restrict__ merge,
    uint32_t *__restrict__ flaggedTiles, int tilesX, int tiles, FusedOut fo, uint32_t *__restrict__ verdictWord, uint32_t *__restrict__ hostWord) {
    // Two uses.  tileFlags == nullptr: the literal kernel for the whole frame (LFG_MOTION_MODE=1), one workgroup per tile of
    // the 2-D grid.  Otherwise the SECOND PASS of the prefiltered path, ONE launch of a 1-D grid whatever the prefilter flagged
    // -- usually nothing: every workgroup reads the count and leaves.  The workgroups share out the ITEMS below, workgroup b
    // taking items b, b + grid, ...: max(tiles, kShareBelow * kFallbackParts) workgroups take one each; a smaller grid -- what a
    // lane launches with frames in flight when its previous call flagged nothing (2,048 workgroups of 42 KB of LDS that only
    // read a count and leave still have to be placed behind the other lanes' kernels: 1.4 % of the frame rate) -- loops, in a
    // kernel of its own (motion_tiled_8_16_loop_kernel):
    //   * up to kShareBelow flagged tiles (counted and listed on the device as they were flagged: flaggedTiles[0], [1 ..]):
    //     a workgroup needs 4.3 ms for a tile whatever else the chip is doing, so each tile is shared by kFallbackParts
    //     workgroups, each on a contiguous part of the tie order.  The parts meet in `merge` (one 64-bit word per pixel
    //     of the slot's 64 x 64 tile, all ones between calls): atomicMin of (cost bits << 32 | rank) is the smallest cost
    //     and, among equal costs, the first candidate in tie order; the part that arrives LAST (a counter per slot behind
    //     the list) turns the words into vectors and leaves them all ones again.  (Round 2: two launches for the two
    //     regimes, a third for the merge, and the words lived in the lists, preset by the resolve kernel.)
    //   * more flagged tiles fill the chip by themselves: one workgroup each, vectors written directly.
    // Candidates are visited in TIE ORDER: rank2scan[r] is the scan index (dy+R)*33 + (dx+R) of the r-th candidate
    // and the first strict minimum wins, so the table decides between equal costs.  Reference semantics: the
    // identity (motion.comp's scan order).  Entry kCand is a sentinel (scan index kCand, one row below the window).
    __shared__ __attribute__((aligned(16))) float sD[2][kDH * kDS];    // 2 x 21 KB
    __shared__ uint32_t sLast;
    int tileX = (int)blockIdx.x, tileY = (int)blockIdx.y;
    int parts = 1, part = 0, slot = 0;
    if (tileFlags) {
        const uint32_t flagged = *flaggedTiles;
        // (for the host, which sizes the lane's NEXT launch of this pass by it: bit 30 of the call's verdict word -- which this
        //  launch, the call's last, also delivers: one store into the host's pinned word, not a copy command behind the call)
        if (verdictWord && blockIdx.x == 0 && threadIdx.x == 0) {
            const uint32_t word = *verdictWord | (flagged != 0u ? 1u << 30 : 0u);
            if (flagged != 0u) *verdictWord = word;
            if (hostWord) __hip_atomic_store(hostWord, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (flagged == 0u) return;                                     // the usual case
        if (!fallback_item((int)blockIdx.x, flagged, tileFlags, flaggedTiles, tilesX, tiles, tileX, tileY, parts, part, slot)) return;
    }
    exact_tile<kFused>(prev, prevPitch, curr, currPitch, mv, mvPitch, W, H, rank2scan, merge, flaggedTiles, fo, tileX, tileY, parts, part, slot, sD, sLast);
}

// The same pass on a grid smaller than its items (see above): workgroup b takes items b, b + grid, ...  A kernel of its own: with
// the loop around it the tile's code keeps more alive (128 registers and a spill where the kernel above has 117), and the kernel
// above is the one that runs where tiles ARE flagged.
template <bool kFused>
__global__ __launch_bounds__(kNT, 4) void motion_tiled_8_16_loop_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H, const uint32_t *__restrict__ tileFlags,
    const uint32_t *__restrict__ rank2scan, unsigned long long *__restrict__ merge,
    uint32_t *__restrict__ flaggedTiles, int tilesX, int tiles, FusedOut fo, uint32_t *__restrict__ verdictWord, uint32_t *__restrict__ hostWord) {
    __shared__ __attribute__((aligned(16))) float sD[2][kDH * kDS];
    __shared__ uint32_t sLast;
    const uint32_t flagged = *flaggedTiles;
    if (verdictWord && blockIdx.x == 0 && threadIdx.x == 0) {          // (see the kernel above)
        const uint32_t word = *verdictWord | (flagged != 0u ? 1u << 30 : 0u);
        if (flagged != 0u) *verdictWord = word;
        if (hostWord) __hip_atomic_store(hostWord, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (flagged == 0u) return;
    const int items = flagged <= (uint32_t)kShareBelow ? kShareBelow * kFallbackParts : tiles;
    for (int item = (int)blockIdx.x; item < items; item += (int)gridDim.x) {
        int tileX, tileY, parts, part, slot;
        if (fallback_item(item, flagged, tileFlags, flaggedTiles, tilesX, tiles, tileX, tileY, parts, part, slot))
            exact_tile<kFused>(prev, prevPitch, curr, currPitch, mv, mvPitch, W, H, rank2scan, merge, flaggedTiles, fo, tileX, tileY, parts, part, slot, sD, sLast);
        __syncthreads();                   // (the next item reuses the distance planes and sLast)
    }
}

// ------------------------------------------------------------------------------ prefiltered path
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

constexpr int kSlabP = 132;                       // slab pitch of a ROW PAIR (floats): rows r, r + 8 interleaved per column,
                                                  // so a thread writes two rows with one ds_write_b64 and the row sums read
                                                  // both with one ds_read_b64.  8-byte accesses are served sixteen lanes at a
                                                  // time over the sixteen 8-byte slots of the 32 banks: with 66 slots per row
                                                  // the sixteen lanes (r8 = 0..7, q = 2k, 2k + 1) of a transposed read fall
                                                  // into slots 2 r8 + 7 q + i mod 16 -- the even ones for one q, the odd ones
                                                  // for the other.  (136 put rows r and r + 4 on the same slots: 5.7e8
                                                  // conflict cycles per launch on a frame that searches in full, measured.)
static_assert((kSlabP / 2) % 16 == 2 && kSlabP / 2 >= 64, "conflict-free transposed 8-byte reads");
#ifndef LFG_SIXTEEN
#define LFG_SIXTEEN 1
#endif
#ifndef LFG_ZERO_COMPARE
#define LFG_ZERO_COMPARE 1
#endif
#ifndef LFG_LIST_MAIN
#define LFG_LIST_MAIN 10
#endif
#ifndef LFG_LIST_AUX
#define LFG_LIST_AUX 10
#endif
#ifndef LFG_LIST_DYN
#define LFG_LIST_DYN 10
#endif
#ifndef LFG_BORDER_PER_SEGMENT
#define LFG_BORDER_PER_SEGMENT 1
#endif
#ifndef LFG_RANK_ALWAYS
#define LFG_RANK_ALWAYS 1
#endif
#ifndef LFG_BAND
#define LFG_BAND 1
#endif
#ifndef LFG_QUEUE_INIT
#define LFG_QUEUE_INIT 1
#endif
#ifndef LFG_LOOKAHEAD
#define LFG_LOOKAHEAD 3                 // candidates per lane of the lookahead (0 or 1: off; 2 until the ranks gave the window offsets by arithmetic)
#endif
#ifndef LFG_SAD_TEST
#define LFG_SAD_TEST 1                   // one-point lattice test by SAD while the wave's largest threshold is small
#endif
#ifndef LFG_SAD_TEST_MAX
#define LFG_SAD_TEST_MAX 8.0f
#endif
#ifndef LFG_EXACT_MATCH
#define LFG_EXACT_MATCH 1                // a hint whose every block position is the same bytes in both frames skips its evaluation
#endif
#ifndef LFG_FIRST_BATCH
#define LFG_FIRST_BATCH 1                // entries of the first batch: the top hint alone (2: with zero motion, as in round 1)
#endif
// Narrow search (prefilter_unit, "Narrow search"): the band of pixel columns that holds a segment's pixels without a
// match, at most sixteen columns wide, searched several candidates per pass.
#ifndef LFG_NARROW
#define LFG_NARROW 1
#endif
#ifndef LFG_NARROW_THR
#define LFG_NARROW_THR 2048.0f                    // a pixel whose threshold is still this large after the hints has no match
                                                  // (below it the sixteen-point test still drops wrong candidates)
#endif
constexpr int kNarrowMax = 16;                    // pixel columns of the band at most
constexpr int kNarrowPitch = 164;                 // slab pitch of a row pair in the narrow passes (floats): = 4 mod 32, so the
                                                  // transposed reads of 16 rows x 2 column groups fall into 32 distinct banks
constexpr int kNarrowQ = 20;                      // ... whose columns are stored four-way interleaved (column c at (c & 3) * 20 + c / 4)
constexpr int kSlabFloats = 8 * kNarrowPitch;     // a wave's slab: 8 row pairs x kSlabP (wide), 8 row pairs x kNarrowPitch (narrow)
static_assert(kSlabFloats >= 8 * kSlabP && 2 * (3 * kNarrowQ + 15) + 1 < kNarrowPitch, "both layouts fit");
#ifndef LFG_HINT_GRID
#define LFG_HINT_GRID 16
#endif
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

#ifdef LFG_MOTION_STAMPS   // diagnostic build (tools/build_variant.sh stamps -DLFG_MOTION_STAMPS): per-wave timing and counts
__device__ unsigned long long gMotionStamps[8192 * 4 * 8];
__device__ unsigned long long gResolveStats[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, ~0ull, 0, 0};
#endif

// order32[e] = the candidate's rank in the tie order (motion_tables) in the low half, its window offset
// (dx+R)*kWinH + (dy+R) in the high half; 32-bit entries so the wave-uniform reads are scalar loads.
#ifndef LFG_PREF_OCC
#define LFG_PREF_OCC 2                 // waves per SIMD: 2 = 256 VGPRs, no spills (168 VGPRs at 3 spill 89 registers to scratch:
                                       // +24 % on matched content, 6 % faster only where every segment searches in full)
#endif
#ifndef LFG_DYN_PARTS
#define LFG_DYN_PARTS 8                 // parts of the candidate order a handed-over segment is searched in, AT MOST (4 or 8): what the scratch is
#endif                                  // laid out for.  A call uses sp.dynParts: eight when one frame runs at a time (two workgroups halve the
                                        // longest units of a launch: occluded 841 -> 890 frames/s, moving objects 983 -> 1,074), four with
                                        // frames in flight, where the sum of all units' times is what counts and a part's staging and setup
                                        // are paid half as often (occluded 1,047 -> 1,082, moving objects 1,272 -> 1,281).
static_assert(LFG_DYN_PARTS == 4 || LFG_DYN_PARTS == 8, "four parts per queue entry, at most eight lists per pixel in the resolve kernel");

// One work unit of the prefilter (see motion_prefilter_kernel below, which hands units to its workgroups).
// `unit` indexes the plan's unit table, or -- fromQueue -- the queue of segments handed over at run time.
template <bool kFused>
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
#ifdef LFG_MOTION_STAMPS
    const unsigned long long stampStart = __builtin_amdgcn_s_memrealtime();
    unsigned stampEvals = 0u, stampBatches = 0u, stampBox = 0u, stampBand = 0u, stampPass = 0u, stampNarrow = 0u, stampThr = 0u, stampThrEnd = 0u, stampFour = 0u;
    unsigned long long stampStaged = 0ull, stampFirst = 0ull, stampLattice = 0ull, stampAhead = 0ull, stampAheadFirst = 0ull, stampPass7 = 0ull;
    unsigned long long phaseLattice = 0ull, phaseSixteen = 0ull, phaseEval = 0ull;     // (LFG_STAMP_PHASES: time in the batch tests, the sixteen-point test, full evaluations)
#endif
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
#ifndef LFG_RANK_ARITH
#define LFG_RANK_ARITH 1                 // by rank, the window offset of a candidate from its rank by arithmetic (see candidateAt)
#endif
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
#ifndef LFG_HEAD
#define LFG_HEAD 2
#endif
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
    // ---- search window: prev(bx0 - R + wx, by0 - R + wy), zero outside the image (texelFetch semantics); a segment
    // unit stages its own 55 rows only.  Staging is a latency chain (load -> LDS store), so everything a thread loads is
    // in flight at once: ten 16-byte loads of four texels each -- the window starts 20 texels left of a tile whose
    // origin is a multiple of 56, so its groups of four are 16-byte aligned and, when the width is a multiple of four,
    // lie inside the image or outside it as a whole.  Other pitches and widths take single texels, thirteen in flight.
    const int stageRow0 = segUnit ? kSeg * seg : 0;
    const int stageRows = segUnit ? kSegD + 2 * kR : kWinH;
    if ((W & 3) == 0 && (prevPitch & 15) == 0 && ((uintptr_t)prev & 15u) == 0u) {
        constexpr int kGroups = (kWinW + 3) / 4;                       // 24 groups per window row (the last holds 3 texels)
        constexpr int kRounds = (kWinH * kGroups + kPNT - 1) / kPNT;   // 10
        uint4 v[kRounds];
        // (the thread number goes through an empty asm: its row / group split is the same for every unit a persistent
        //  workgroup takes, and hoisted out of that loop the twenty values sit in scratch -- each reload, with the
        //  s_waitcnt vmcnt(0) it brings, then waited for the previous window load: ten loads one after the other)
        int tidL = tid;
        asm volatile("" : "+v"(tidL));
        // (a window that lies inside prev -- nine tiles in ten -- needs neither the clamps nor the masks: 150 instructions
        //  per thread in front of its loads)
        const int stageInside = __builtin_amdgcn_readfirstlane((int)((bx0 - kR >= 0) & (bx0 - kR + 4 * kGroups <= W) &
                                                                      (by0 - kR + stageRow0 >= 0) & (by0 - kR + stageRow0 + stageRows <= H)));
        if (stageInside) {
            // (row and group of round k from those of round k - 1: 256 threads are ten rows of 24 groups and 16 groups more --
            //  one division per unit instead of ten)
            static_assert(kPNT == 10 * kGroups + 16, "a round advances a thread by ten rows and sixteen groups");
            int wyK = tidL / kGroups, gK = tidL - wyK * kGroups;
#pragma unroll
            for (int k = 0; k < kRounds; ++k) {
                const int wy = min(stageRow0 + wyK, stageRow0 + stageRows - 1);   // (the last round's spare threads re-read the last row)
                const int gx = bx0 - kR + 4 * gK, gy = by0 - kR + wy;
                v[k] = *reinterpret_cast<const uint4 *>(prev + (size_t)gy * (size_t)prevPitch + (size_t)gx * 4u);
                gK += 16; wyK += 10;
                if (gK >= kGroups) { gK -= kGroups; wyK += 1; }
            }
        } else {
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int i = k * kPNT + tidL;
            const int wy = stageRow0 + i / kGroups, g = i % kGroups;
            const int gx = bx0 - kR + 4 * g, gy = by0 - kR + wy;
            // branch-free: the load goes to the nearest group inside the image and is dropped afterwards
            const uint4 t = *reinterpret_cast<const uint4 *>(prev + (size_t)clampi(gy, 0, H - 1) * (size_t)prevPitch +
                                                             (size_t)clampi(gx, 0, W - 4) * 4u);
            // (a mask, not a select: the compiler turns the select into a branch that waits for this load before the next
            //  one is issued -- ten memory latencies in a row for every tile that touches the image border)
            const uint32_t keep = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? 0xFFFFFFFFu : 0u;
            v[k] = uint4{t.x & keep, t.y & keep, t.z & keep, t.w & keep};
        }
        }
        asm volatile("" : "+v"(tidL));
        int wyS = tidL / kGroups, gS = tidL - wyS * kGroups;
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int wy = stageRow0 + wyS, g = gS;
            const bool staged = wyS < stageRows;                       // (i < stageRows * kGroups)
            gS += 16; wyS += 10;
            if (gS >= kGroups) { gS -= kGroups; wyS += 1; }
            if (staged) {
                uint32_t *dst = sWin + (4 * g) * kWinH + wy;
                dst[0] = v[k].x; dst[kWinH] = v[k].y; dst[2 * kWinH] = v[k].z;
                if (4 * g + 3 < kWinW) dst[3 * kWinH] = v[k].w;
            }
        }
    } else {
        constexpr int kStageAhead = 13;
        const int stageTexels = stageRows * kWinW;
        for (int r0 = 0; r0 * kPNT < stageTexels; r0 += kStageAhead) {
            uint32_t v[kStageAhead];
#pragma unroll
            for (int k = 0; k < kStageAhead; ++k) {
                const int i = (r0 + k) * kPNT + tid;
                const int wy = stageRow0 + i / kWinW, wx = i % kWinW;  // global reads stay row-major (coalesced)
                const int gx = bx0 - kR + wx, gy = by0 - kR + wy;
                const uint32_t t = *reinterpret_cast<const uint32_t *>(prev + (size_t)clampi(gy, 0, H - 1) * (size_t)prevPitch +
                                                                       (size_t)clampi(gx, 0, W - 1) * 4u);
                v[k] = t & ((gx >= 0 && gx < W && gy >= 0 && gy < H) ? 0xFFFFFFFFu : 0u);      // (a mask: see above)
            }
#pragma unroll
            for (int k = 0; k < kStageAhead; ++k) {
                const int i = (r0 + k) * kPNT + tid;
                const int wy = stageRow0 + i / kWinW, wx = i % kWinW;
                if (i < stageTexels) sWin[wx * kWinH + wy] = v[k];
            }
        }
    }

#pragma unroll
    for (int j = 0; j < kSegD; ++j) cc[j] = __builtin_amdgcn_udot4(c[j], c[j], 0x4B000000u, false);   // 2^23 + |c|^2 as float bits
    // Some candidate's block can leave prev altogether only if the search window does.
    const int windowLeavesPrev = __builtin_amdgcn_readfirstlane(
        !((bx0 - kR >= 0) && (bx0 + kPTW + kB - 2 + kR < W) && (by0 - kR >= 0) && (by0 + kPTH + kB - 2 + kR < H)));

    __syncthreads();                       // window staged; the only workgroup barrier
#ifdef LFG_MOTION_STAMPS
    stampStaged = __builtin_amdgcn_s_memrealtime();
#endif
    if (ty0 + kSeg * seg >= H) return;     // this wave's rows lie below the image
    if (whole && !fromQueue && __builtin_amdgcn_readfirstlane((int)segDone[tile * (kPTH / kSeg) + seg]) != 0) return;     // (settled by the lean kernel)

    // (lane 63 has no position column: it re-reads lane 62's texels, its sums are never used)
    const lds_ro_u32_ptr winBase = (lds_ro_u32_ptr)(sWin + min(lane, kPTW + kB - 2) * kWinH + kSeg * seg);
    auto fetchWindow = [&](uint32_t (&p)[kSegD], uint32_t ord) {
        const lds_ro_u32_ptr w = winBase + (ord >> 16);
#pragma unroll
        for (int j = 0; j < kSegD; ++j) p[j] = w[j];
    };
    // Packed fp32 throughout (v_pk_add_f32: two adds per issue slot; a wave issues one VALU op per four
    // cycles whatever its width): position j is paired with position j + 8 -- A[j] = (d_j, d_{j+8}), j = 0 .. 14, the
    // distances of positions 8 .. 14 sitting in two pairs -- so every level of the sliding tree is "pair j + pair j+k",
    // nothing is left over for scalar adds, and the tree ends in C8[r] = (V8 of row r, V8 of row r + 8): the very pairs
    // the slab takes (one ds_write_b64 each) and the row sums want (transpose).  (Round 2 paired j with j + 12: its
    // outputs came as (0,12) .. (3,15) plus eight scalars and were shuffled into row pairs with 27 moves per candidate.)
    constexpr int kPairs = kSegD - 8;                                  // 15
    auto columnSums = [&](const uint32_t (&p)[kSegD], const uint32_t (&c)[kSegD], const uint32_t (&cc)[kSegD], const uint32_t valid,
                          f32x2 (&C8)[8]) {
        // n = |c|^2 + |p|^2 - 2 c.p without integer->float conversions or shifts (half-rate ops on gfx950):
        // the dot products accumulate onto float bit patterns, 0x4B000000 + k = 2^23 + k and
        // 0x4B800000 + k = 2^24 + 2k (k < 2^23), so two exact fp32 operations give n as a float.
        // (No inline asm on dot results: a VALU op that reads a v_dot4 result needs 3 wait states on
        //  gfx950, which only the compiler's hazard recogniser provides.)
        auto f1of = [&](int j) { return __builtin_bit_cast(float, __builtin_amdgcn_udot4(p[j], p[j], cc[j], false)); };        // 2^23 + |c|^2 + |p|^2
        auto f2of = [&](int j) { return __builtin_bit_cast(float, __builtin_amdgcn_udot4(c[j], p[j], 0x4B800000u, false)); };  // 2^24 + 2 c.p
        const f32x2 kBias = {8388608.0f, 8388608.0f};
        f32x2 A[kPairs];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                  // positions 0 .. 15
            const f32x2 F1 = {f1of(j), f1of(j + 8)}, F2 = {f2of(j), f2of(j + 8)};
            const f32x2 N = (F1 - F2) + kBias;
            A[j] = f32x2{__builtin_amdgcn_sqrtf(N.x), __builtin_amdgcn_sqrtf(N.y)};
        }
        float dHi[kSegD - 16];                                         // positions 16 .. 22
#pragma unroll
        for (int k = 0; k + 1 < kSegD - 16; k += 2) {
            const f32x2 F1 = {f1of(16 + k), f1of(17 + k)}, F2 = {f2of(16 + k), f2of(17 + k)};
            const f32x2 N = (F1 - F2) + kBias;
            dHi[k] = __builtin_amdgcn_sqrtf(N.x); dHi[k + 1] = __builtin_amdgcn_sqrtf(N.y);
        }
        dHi[kSegD - 17] = __builtin_amdgcn_sqrtf((f1of(kSegD - 1) - f2of(kSegD - 1)) + 8388608.0f);
        // Border segments only: a REAL wave-uniform branch (the flag is laundered through an empty asm so the loop is
        // not unswitched into two copies, which doubles the register pressure of the function, and the asm inside the
        // body keeps the compiler from turning the branch into 23 selects -- with their 23 lane masks, half of them
        // reloaded from spilled scalars -- that every evaluation of every interior segment then executes).
        int border = borderSeg;
        asm volatile("" : "+s"(border));
        if (border) {                      // position outside the image: skipped by the shader, adds 0 here
            asm volatile("; positions outside the image");
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                A[j].x = ((valid >> j) & 1u) ? A[j].x : 0.0f;
                A[j].y = ((valid >> (j + 8)) & 1u) ? A[j].y : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < kSegD - 16; ++k) dHi[k] = ((valid >> (16 + k)) & 1u) ? dHi[k] : 0.0f;
        }
#pragma unroll
        for (int j = 8; j < kPairs; ++j) A[j] = f32x2{A[j - 8].y, dHi[j - 8]};
        f32x2 B[kPairs - 1], G[kPairs - 3];
#pragma unroll
        for (int j = 0; j < kPairs - 1; ++j) B[j] = A[j] + A[j + 1];   // (v2_j, v2_{j+8}), j = 0..13
#pragma unroll
        for (int j = 0; j < kPairs - 3; ++j) G[j] = B[j] + B[j + 2];   // (v4_j, v4_{j+8}), j = 0..11
#pragma unroll
        for (int j = 0; j < 8; ++j) C8[j] = G[j] + G[j + 4];           // (v8_j, v8_{j+8}), j = 0..7
    };

    // ---- row sums: rows r8 and r8 + 8 of the wave's 16, pixels tx0 + 7 q .. + 6
    const int r8 = lane & 7, q = lane >> 3;
    const int px0 = tx0 + kRun * q;
    typedef const __attribute__((address_space(3))) f32x2 *lds_ro_f32x2_ptr;
    f32x2 *const slabW = reinterpret_cast<f32x2 *>(sSlab[wave]) + lane;               // write: rows (r, r + 8), column lane
    const lds_ro_f32x2_ptr slabR = (lds_ro_f32x2_ptr)(sSlab[wave]) + r8 * (kSlabP / 2) + kRun * q;     // read: rows (r8, r8 + 8), input column 7 q + i
    f32x2 thr2[kRun];                     // (row r8, row r8 + 8): kRatio * (smallest S~ so far); a candidate survives
                                          // while S~ <= thr
    uint32_t cnt2[2][(kRun + 1) / 2];                                  // 16-bit counters (1089 events at most)
    // Records {S~ bits, candidate}: record k of pixel (x, y) lives at list[(y * kListK + k) * W + x] -- K planes
    // per image row, so the resolve kernel reads record k of 64 neighbouring pixels as one 512-byte line.
    // Address = wave-uniform base + 32-bit lane offset; k * W is a 24-bit multiply.
    // Whole tiles write into the image-shaped arrays (row stride W), shared tiles into their unit's private
    // 56 x 64 block of the auxiliary arrays (row stride 56).
    // (auxiliary blocks: a tile's 64 rows for the units of the plan, a segment's 16 rows for queued units)
    const int auxUnit = whole ? 0 : (fromQueue ? 4 * unit + wave : (int)unitAuxFirst + chunk);
    const int auxRows = fromQueue ? kSeg : kPTH, auxRow0 = fromQueue ? kSeg * seg : 0;
    Rec *const auxListBase = fromQueue ? sp.dynList : sp.auxList;
    float *const auxUminBase = fromQueue ? sp.dynUmin : sp.auxUmin;
    uint32_t *const auxCountBase = fromQueue ? sp.dynCount : sp.auxCount;
    const uint32_t rowStride = whole ? (uint32_t)W : (uint32_t)kPTW;
    const uint32_t listK = whole ? (uint32_t)kListK : (fromQueue ? (uint32_t)kListDyn : (uint32_t)kListAux);      // this unit's list depth
    Rec *const waveList = whole
        ? list + ((size_t)(ty0 + kSeg * seg) * (size_t)kListK * (size_t)W + (size_t)tx0)
        : auxListBase + ((size_t)auxUnit * auxRows + (size_t)(kSeg * seg - auxRow0)) * (size_t)listK * (size_t)kPTW;
    uint32_t laneOff[2];
    // (a segment whose 16 x 56 pixels all lie inside the image -- all but the last column and row of tiles -- skips the
    //  fourteen per-pixel tests: their lane masks were kept in scalar registers, spilled, across the whole unit)
    const int allInside = __builtin_amdgcn_readfirstlane((int)((tx0 + kPTW <= W) & (ty0 + kSeg * seg + kSeg <= H)));
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
        laneOff[hb] = (uint32_t)(8 * hb + r8) * (listK * rowStride) + (uint32_t)(kRun * q);   // < 16 K W
#pragma unroll
        for (int i = 0; i < (kRun + 1) / 2; ++i) cnt2[hb][i] = 0u;
    }
    if (allInside) {
#pragma unroll
        for (int i = 0; i < kRun; ++i) thr2[i] = f32x2{__builtin_inff(), __builtin_inff()};
    } else {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int py = ty0 + kSeg * seg + 8 * hb + r8;
#pragma unroll
            for (int i = 0; i < kRun; ++i)    // pixels outside the image never pass the test (S~ >= 0 > -inf)
            {
                const float t0 = (py < H && px0 + i < W) ? __builtin_inff() : -__builtin_inff();
                if (hb) thr2[i].y = t0; else thr2[i].x = t0;
            }
        }
    }
    // some pixel of this lane holds listK records or more: bit 15 of a 16-bit counter biased by 0x8000 - listK.  (The last
    // slot is a spare -- candidates that are written without being counted, plateau duplicates and the head of the order
    // in units that share a tile, go to slot `count` and must not land on a record -- so a list holds listK - 1.)
    auto listsOverflowed = [&]() -> bool {
        uint32_t any = 0u;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < (kRun + 1) / 2; ++i) any |= cnt2[hb][i] + (0x8000u - listK) * 0x10001u;
        }
        return (any & 0x80008000u) != 0u;
    };
    // Row sums of both runs of a lane at once: X[i] = (row r8, row r8 + 8) at input column i, so every level of
    // the sliding tree is one packed add per entry (13 + 11 + 7) and s2[i] = (S~ of pixel i in row r8, in row r8+8).
    auto runSums = [&](const f32x2 (&X)[kRunIn], f32x2 (&s2)[kRun]) {
        f32x2 h2[kRunIn - 1], h4[kRunIn - 3];
#pragma unroll
        for (int i = 0; i < kRunIn - 1; ++i) h2[i] = X[i] + X[i + 1];
#pragma unroll
        for (int i = 0; i < kRunIn - 3; ++i) h4[i] = h2[i] + h2[i + 2];
#pragma unroll
        for (int i = 0; i < kRun; ++i) s2[i] = h4[i] + h4[i + 4];
    };
    // The test has levels: a packed "does anything pass" check (level 0, below), then all fourteen comparisons
    // (VALU -> scalar masks, pipelined) and one SCALAR branch per pixel; only a taken branch touches EXEC.  A divergent
    // `if` per pixel costs ~6 VALU slots each in compare -> saveexec -> branch latency (tools/bench_intops.hip).
    uint32_t plateauSeen = 0u;            // bit 7 hb + i: this pixel's list already holds a member of its plateau
    auto rowSumsAndTest = [&](const f32x2 (&X)[kRunIn], uint32_t ord, uint32_t countIt) {
        const uint32_t cand = ord & 0xFFFFu;
        const int candDx = (int)((ord >> 16) / (uint32_t)kWinH) - kR, candDy = (int)((ord >> 16) % (uint32_t)kWinH) - kR;
        const uint32_t zeroCap = 0x00800000u + cand;                   // float bits, wave-uniform
        // can this candidate's block leave prev for ANY pixel of the segment?  (wave-uniform; next to the rim a third of
        // the candidates can, and the per-pixel plateau bookkeeping below is most of what a recorded candidate costs)
        const int segY0 = ty0 + kSeg * seg;
        const bool candMayLeave = windowLeavesPrev &&
            block_leaves_prev_any(tx0, min(tx0 + kPTW - 1, W - 1), segY0, min(segY0 + kSeg - 1, H - 1), candDx, candDy, W, H);
        f32x2 s2[kRun];
        runSums(X, s2);
        // Level 0: does ANY of the lane's 14 pixels pass (thr - S~ >= 0 for some of them)?  Seven packed
        // subtractions and a max tree instead of fourteen compares and scalar branches; with the hints in front,
        // most candidates end here.  (inf - S~ = inf passes, -inf never does; a difference flushed to zero can
        // only send us into the exact per-pixel test below for nothing.)
        f32x2 dm = thr2[0] - s2[0];
        float top = __builtin_fmaxf(dm.x, dm.y);
#pragma unroll
        for (int i = 1; i < kRun; ++i) {
            dm = thr2[i] - s2[i];
            top = __builtin_fmaxf(top, __builtin_fmaxf(dm.x, dm.y));                     // v_max3_f32
        }
        if (__builtin_amdgcn_readfirstlane(__ballot(top >= 0.0f) == 0ull)) return;
#ifdef LFG_DIAG_NO_SLOW   // timing experiment (wrong results): thresholds follow the minima, nothing is recorded
#pragma unroll
        for (int i = 0; i < kRun; ++i) {
            const f32x2 capped = s2[i] * f32x2{kRatio, kRatio};
            thr2[i].x = __builtin_fminf(thr2[i].x, capped.x); thr2[i].y = __builtin_fminf(thr2[i].y, capped.y);
        }
        return;
#endif
        float sv[2][kRun], thr[2][kRun];
#pragma unroll
        for (int i = 0; i < kRun; ++i) { sv[0][i] = s2[i].x; sv[1][i] = s2[i].y; thr[0][i] = thr2[i].x; thr[1][i] = thr2[i].y; }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            // seven masks at a time (all fourteen at once is more scalar registers than the loop has to spare)
            unsigned long long hit[kRun];
#pragma unroll
            for (int i = 0; i < kRun; ++i) hit[i] = __ballot(sv[hb][i] <= thr[hb][i]);
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                if (hit[i] != 0ull) {                                  // wave-uniform
                    asm volatile("; some lane records a candidate");  // keeps this a scalar branch of its own
                    const float s = sv[hb][i];
                    if (s <= thr[hb][i]) {                             // rare: ~7.6 times per pixel in 1089
                        // S~ == 0 means every distance is exactly 0, so the shader's cost is exactly 0 too: the
                        // answer is the first such candidate in scan order.  Its index goes into the threshold
                        // itself (zeroCap = 0x00800000 + cand as a float: tiny, yet above 0 and below every
                        // non-zero S~ >= 1) and the list does not grow -- static or flat areas, where many
                        // candidates cost 0, never fill the lists.  max() picks zeroCap only for S~ == 0.
                        float cap, t;
                        asm("v_max_f32 %0, %2, %1" : "=v"(cap) : "v"(s * kRatio), "s"(zeroCap));   // no NaNs here
                        asm("v_min_f32 %0, %1, %2" : "=v"(t) : "v"(thr[hb][i]), "v"(cap));
                        if (hb) thr2[i].y = t; else thr2[i].x = t;
                        // `restart`: the candidate undercuts the old threshold by more than the bracket's width, so every
                        // earlier record of this pixel is dead (see the list depths): the count starts again at 0
                        const bool restart = s < thr[hb][i] * kRestart;
                        if (restart) cnt2[hb][i >> 1] &= ~(0xFFFFu << (16 * (i & 1)));
                        const uint32_t n = (cnt2[hb][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                        // past the end of the list the last slot is overwritten; the count keeps growing and
                        // flags the tile
                        const uint32_t at = __umul24(min(n, listK - 1u), rowStride) + laneOff[hb] + (uint32_t)i;
                        if (s != 0.0f) waveList[at] = rec_make(s, cand);       // (a zero-cost candidate lives in the threshold word: no record, no write)
                        uint32_t inc = (s != 0.0f && countIt != 0u) ? (1u << (16 * (i & 1))) : 0u;
                        if (candMayLeave) {                            // wave-uniform: tiles away from the rim skip this
                            // one member per plateau (block_leaves_prev): a second one is written but not counted
                            // (the pixel's coordinates are recomputed from the lane number here: kept in registers
                            //  across the evaluation they end up spilled, and a scratch reload costs a memory latency)
                            uint32_t l;
                            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l));
                            const uint32_t bit = 1u << (kRun * hb + i);
                            const bool plateau = block_leaves_prev(tx0 + kRun * (int)(l >> 3) + i, ty0 + kSeg * seg + 8 * hb + (int)(l & 7u),
                                                                   candDx, candDy, W, H);
                            inc = (plateau && (plateauSeen & bit) != 0u) ? 0u : inc;
                            plateauSeen |= (plateau && inc != 0u) ? bit : 0u;
                        }
                        cnt2[hb][i >> 1] += inc;
                    }
                }
            }
        }
    };
    // slab traffic of one candidate: the sixteen rows out as eight (row r, row r + 8) pairs per column, the transposed runs in as
    // thirteen pairs per lane -- ONE round trip (round 2: rows 0-7 out and in, then rows 8-15; 26 reads).  Conflict-free by the
    // pitch (kSlabP); the writes are consecutive.
    // (The wave-scope fences are for the compiler only: nothing else tells it that the reads of a pass must stay between
    //  that pass's writes and the next pass's.  The hardware executes one wave's DS operations in order.)
    auto transpose = [&](const f32x2 (&C8)[8], f32x2 (&X)[kRunIn], const lds_ro_f32x2_ptr from) {
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) slabW[r * (kSlabP / 2)] = C8[r];
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < kRunIn; ++i) X[i] = from[i];
        wave_lds_sync();
    };

    bool settledAtZero = false;            // set by run(): the wave's largest threshold stands for a zero cost
    auto run = [&]() -> int {              // 0: done, 1: lists overflowed (tile flagged), 2: segment handed over
#ifdef LFG_STAMP_LATTICE4
        unsigned long long stampU[4] = {0ull, 0ull, 0ull, 0ull};
        stampU[0] = __builtin_amdgcn_s_memrealtime();
#endif
        uint32_t p[kSegD];
        f32x2 v8[8];                       // (V8 of row r, of row r + 8)
        f32x2 x[kRunIn];
        // Once EVERY pixel of this wave owns a zero-cost candidate, a candidate can only still matter if it comes
        // earlier in the tie order than the latest of those: zeroBound = that rank (none: 0xFFFFFFFF).  Later-ranked
        // candidates are skipped outright -- static areas and exact translations end the search early, exactly as
        // "stop at cost 0" would, without changing any result.
        uint32_t zeroBound = 0xFFFFFFFFu;
        float waveThr = __builtin_inff();
        auto refreshZeroBound = [&]() {
            uint32_t k = 0u;
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                // (through named floats: __builtin_bit_cast applied to the vector-element lvalue thr2[i].y yields
                //  the bits of .x with this compiler)
                const float fx = thr2[i].x, fy = thr2[i].y;
                const uint32_t b0 = __builtin_bit_cast(uint32_t, fx), b1 = __builtin_bit_cast(uint32_t, fy);
                k = max(k, b0 == 0xFF800000u ? 0u : b0);               // -inf: a pixel outside the image
                k = max(k, b1 == 0xFF800000u ? 0u : b1);
            }
            k = wave_max_u32(k);
            zeroBound = k < 0x00800000u + (uint32_t)kCand ? (k >= 0x00800000u ? k - 0x00800000u : 0u) : 0xFFFFFFFFu;
            zeroBound = (uint32_t)__builtin_amdgcn_readfirstlane((int)zeroBound);
            waveThr = __builtin_bit_cast(float, k);                    // largest threshold of the wave's pixels (+inf: none yet)
        };
        // Partial-distortion test.  S~ is a sum of non-negative distances, and a rounded fp32 sum of non-negative
        // terms is never below any of its terms, so S~(p, m) >= D_m(c) for every position c of p's block.  A lattice
        // of positions that puts a point into every pixel's 8 x 8 block therefore certifies the whole wave: if the
        // candidate's distance exceeds waveThr at all of them, no pixel can pass its own (smaller) threshold and the
        // candidate is dropped after a handful of its 23 x 63 distances.
        //   interior tiles: rows 7|15, columns = 7 mod 8 -- exactly one point per block, 2 x 7 in all (kLatC0 ...);
        //   border tiles:   rows 4|8|12|16, columns = 0 mod 4 in 4..56 -- the point (4*floor(x/4), 4*floor(y/4)) of
        //                   pixel (x, y) lies in its block and inside the image whenever the pixel does, so
        //                   positions outside the image are simply left out; 4 x 14 points.
        // A wave tests a BATCH of up to 64 candidates at once, one per lane: the lane fetches its candidate's entry of
        // the order from LDS and walks the lattice -- per point one texel of the window (the candidate's offset, the
        // point's position as an immediate), the current-frame texel of the point from the lane that owns it
        // (v_readlane: wave-uniform), one distance, one running minimum.  All chains are independent, so the batch
        // runs at issue rate, and one ballot gives the candidates that survive.  The thresholds a batch is tested
        // against are those at its start; the first two batches are short (zero motion and the first hint, then six
        // more) so that the long ones start with thresholds worth testing against.
        const int nEntries = nHead + eEnd - eBegin;                    // the head, then this wave's part of the order
        int border = borderSeg;
        asm volatile("" : "+s"(border));   // one copy of the loop below, not one per kind of tile
        // The SQUARED distance n (an exact integer, same arithmetic as columnSums) is enough here: the lattice test
        // compares with a squared threshold and saves the square root, the most expensive operation of a point.
        auto distanceOf = [&](uint32_t cT, uint32_t texel) {
            const uint32_t ccT = __builtin_amdgcn_udot4(cT, cT, 0x4B000000u, false);
            const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(texel, texel, ccT, false));
            const float f2 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(cT, texel, 0x4B800000u, false));
            return (f1 - f2) + 8388608.0f;
        };
#ifndef LFG_ONEPOINT_MAX
#define LFG_ONEPOINT_MAX 32.0f
#endif
        constexpr float kOnePointMax = LFG_ONEPOINT_MAX;   // below: the one-point test alone (cheap, and strong while thresholds are small)
        // "Some lattice point of the candidate is the same four bytes in both frames": what a pixel that already owns a
        // zero-cost candidate needs before another candidate can matter to it (its S~ would have to be 0: every distance
        // of its block exactly 0, the lattice point inside the block among them).  One compare per point.
        auto zeroHit = [&](const uint32_t ordL) -> bool {
            const lds_ro_u32_ptr w = (lds_ro_u32_ptr)(sWin + kSeg * seg) + (ordL >> 16);
            // (the smallest XOR over the points is 0: VALU only -- see the lookahead of the batch loop)
            uint32_t acc = 0xFFFFFFFFu;
            if (border) {
                static_assert(kPTW / 4 == 14, "two halves of seven lattice columns");
#pragma unroll
                for (int half = 0; half < 2; ++half) {                 // (28 reads in flight at once: two LDS round trips, not fourteen)
                    uint32_t tex[7][4];
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
#pragma unroll
                        for (int row = 4; row <= 16; row += 4) tex[k][row / 4 - 1] = w[(4 + 4 * (7 * half + k)) * kWinH + row];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 7; ++k) {
                        const int col = 4 + 4 * (7 * half + k);
                        const uint32_t inImage = (uint32_t)__builtin_amdgcn_readlane((int)valid, col);
#pragma unroll
                        for (int row = 4; row <= 16; row += 4) {
                            const uint32_t drop = ((inImage >> row) & 1u) - 1u;          // a point outside the image is no point (scalar: 0 or all ones)
                            acc = min(acc, (tex[k][row / 4 - 1] ^ (uint32_t)__builtin_amdgcn_readlane((int)c[row], col)) | drop);
                        }
                    }
                }
            } else {
                uint32_t tex[kLatCols][kLatRows];
#pragma unroll
                for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                    for (int t = 0; t < kLatRows; ++t) tex[ci][t] = w[(kLatC0 + 8 * ci) * kWinH + kLatR0 + 8 * t];
                }
#pragma unroll
                for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                    for (int t = 0; t < kLatRows; ++t)
                        acc = min(acc, tex[ci][t] ^ (uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci));
                }
            }
            const bool same = acc == 0u;
            return same;
        };
        // THE BAND.  Once the hints have been tried, the pixels that do not own a zero-cost candidate yet often lie in a few
        // columns -- next to the left and right border the upscaler filters a shifted frame differently, so the match is
        // only nearly exact there, costs a few hundred, and the four- and sixteen-point tests run for every candidate of
        // the segment (a fifth of the prefilter's time on a pan).  Their lattice walks then cover only the groups that hold
        // a pixel of the band [bandLo, bandHi] (pixel columns of the tile); every other pixel is settled at zero cost and
        // is answered for by zeroHit.  (The set of unsettled pixels only shrinks: the band stays valid once computed.)
        int bandLo = 0, bandHi = kPTW - 1;
        auto computeBand = [&]() {
            uint32_t lo = 0u, hi = 0u;                                 // kPTW - first column, last column + 1 (0: none)
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const float fx = thr2[i].x, fy = thr2[i].y;
                const int b0 = __builtin_bit_cast(int, fx), b1 = __builtin_bit_cast(int, fy);     // (-inf is negative: outside the image or narrow)
                if (b0 >= 0x3F000000 || b1 >= 0x3F000000) { lo = max(lo, (uint32_t)(kPTW - (kRun * q + i))); hi = max(hi, (uint32_t)(kRun * q + i + 1)); }
            }
            lo = wave_max_u32(lo); hi = wave_max_u32(hi);
            lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo); hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)hi);
            if (hi != 0u) { bandLo = kPTW - (int)lo; bandHi = (int)hi - 1; }
#ifdef LFG_MOTION_STAMPS
            stampBand = 0x10000u | (unsigned)bandLo | ((unsigned)bandHi << 8);
#endif
        };
#ifndef LFG_FOURPOINT_MAX
#define LFG_FOURPOINT_MAX (4.0f * 510.0f)
#endif
        bool useFourPoint = true, useSixteen = LFG_SIXTEEN != 0, useEight = true;
        auto fourPointApplies = [&]() { return useFourPoint && !(waveThr < kOnePointMax) && waveThr < LFG_FOURPOINT_MAX; };
        // (per lane: the candidate `ordL` still has to be evaluated in full)
        // (Tried in round 3 and dropped: the same walk on sums of absolute differences first -- a distance is at least half
        //  its SAD, one v_sad_u8 per point instead of three dot products, two adds and a square root; it clears 98.3 % of
        //  the candidates for a whole wave on frames with sensor noise where the distances clear 99.6 %, its survivors
        //  deferred to the test by distances -- +1 % on those frames with frames in flight, -2 % one at a time and on a pan.)
        auto fourPointBatch = [&](const uint32_t ordL, bool need, const bool fullBatch) -> bool {
                // FOUR-point test, for thresholds a single distance rarely exceeds (a match under sensor noise costs a
                // few hundred).  Block positions = 3 mod 4 form a lattice of 5 x 15 points of which every pixel's
                // 8 x 8 block holds exactly a 2 x 2 group (columns 3 + 4 g, 7 + 4 g with g = i / 4 for pixel column i,
                // rows likewise: the block starts at the pixel's own position), and S~ >= (the sum of those four
                // distances) (1 - 8.1 u): S~ adds the same rounded distances in a depth-6 tree (factor (1-u)^6 at
                // worst), the group sum in two levels ((1+u)^2 at most).  If every group's sum exceeds
                // waveThr (1 + 2^-19), no pixel passes.  Positions outside the image add nothing to S~ and nothing
                // here; groups that no pixel inside the image maps to are left out.  Walks the lattice column by
                // column: five distances, four vertical pair sums, four group sums with the previous column's.
                // (Positions = 0 mod 4 did the same with 6 x 16 points until late in round 2.)
                auto bitsOf = [](float d) { return __builtin_bit_cast(uint32_t, d); };
                int tx0L = tx0, rowsL = ty0 + kSeg * seg;
                asm volatile("" : "+s"(tx0L), "+s"(rowsL));
                uint32_t rowRelevant = 0u;                              // bit h: some pixel row inside the image maps to vertical group h (rows 4 h .. 4 h + 3)
#pragma unroll
                for (int h = 0; h < 4; ++h) rowRelevant |= (!border || rowsL + 4 * h < H) ? (1u << h) : 0u;
                // (a rolled loop, three columns per trip -- fifteen columns: two per trip leave a remainder whose extra copy cost
                //  11 spilled VGPRs; all of them unrolled were 2,000 instructions and 7 % slower where this test dominates, not
                //  for instruction-cache misses, SQC_ICACHE_MISSES stays at 0.01 %)
                const int gLo = bandLo >> 2, gHi = bandHi >> 2;         // groups i / 4 of the band's pixel columns i
                const bool restricted = LFG_BAND && (bandLo > 0 || bandHi < kPTW - 1);
                // (four copies of the walk: with and without the band's bounds -- the bounds' tests in the one walk cost the
                //  segments whose band is the whole width, noise everywhere, 6 % -- and with and without the masks of a segment
                //  at the image border: a select per point and per group is a third of the SAD walk's instructions)
                // ... times two currencies (round 4): by DISTANCES, as described, and by sums of absolute differences -- a distance is
                // at least half the sum of its four absolute channel differences (Cauchy-Schwarz), so S~ >= (the four SADs' sum) / 2
                // (1 - 10 u) as well, one v_sad_u8 and integer adds per point where a distance is three dot products, two adds and a
                // square root.  Weaker by the ratio of the two norms (0.75 - 0.87 on natural differences): used while the wave's
                // largest threshold is below LFG_FOUR_SAD_MAX -- a match under sensor noise -- where a wrong candidate's four SADs are
                // still twice the threshold; what it lets through waits for the eight- and sixteen-point walks with the rest.
                auto walk = [&](auto banded, auto atBorder, auto sadTag) -> uint32_t {
                constexpr bool kBanded = decltype(banded)::value, kBorder = decltype(atBorder)::value, kSad = decltype(sadTag)::value;
                typedef typename std::conditional<kSad, uint32_t, float>::type T;
                auto asBits = [&](T v) -> uint32_t { if constexpr (kSad) return (uint32_t)v; else return __builtin_bit_cast(uint32_t, v); };
                lds_ro_u32_ptr w = (lds_ro_u32_ptr)(sWin + kSeg * seg) + (ordL >> 16) + (3 * kWinH + 3);
                uint32_t pMin = 0x7F800000u;
                // a group that is not wanted -- the one "before" the first lattice column -- is kept out of the minimum by a
                // previous column whose sums are huge, not by a select per group
                T vPrev[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) vPrev[h] = kBorder ? (T)0 : (kSad ? (T)0x10000000u : (T)1.0e30f);
#pragma unroll 3
                for (int a = 0; a < kPTW / 4 + 1; ++a, w += 4 * kWinH) {   // lattice column 3 + 4 a closes group a - 1
                    if (kBanded && (a < gLo || a > gHi + 1)) continue;  // (wave-uniform)
                    // (the lane select goes through an empty asm: the current-frame texels are the same in every batch,
                    //  and hoisted out of the batch loop their scalar registers are spilled)
                    int colL = 3 + 4 * a;
                    asm volatile("" : "+s"(colL));
                    uint32_t tex[5];
#pragma unroll
                    for (int b = 0; b < 5; ++b) tex[b] = w[4 * b];
                    const uint32_t inImage = kBorder ? (uint32_t)__builtin_amdgcn_readlane((int)valid, colL) : 0xFFFFFFFFu;
                    // group (a - 1, h): its pixels are columns 4 (a - 1) .. + 3, rows 4 h .. + 3
                    const bool colRelevant = a > (kBanded ? gLo : 0) && (!kBorder || tx0L + 4 * (a - 1) < W);
                    T d[5];
#pragma unroll
                    for (int b = 0; b < 5; ++b) {
                        const uint32_t cT = (uint32_t)__builtin_amdgcn_readlane((int)c[3 + 4 * b], colL);
                        const uint32_t keep = 0u - ((inImage >> (3 + 4 * b)) & 1u);    // scalar: all ones or zero
                        if constexpr (kSad) {
                            const uint32_t dd = __builtin_amdgcn_sad_u8(cT, tex[b], 0u);
                            d[b] = kBorder ? (dd & keep) : dd;
                        } else {
                            const float dd = __builtin_amdgcn_sqrtf(distanceOf(cT, tex[b]));
                            d[b] = kBorder ? __builtin_bit_cast(float, bitsOf(dd) & keep) : dd;
                        }
                    }
                    T v[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) v[h] = d[h] + d[h + 1];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const uint32_t g = asBits(vPrev[h] + v[h]);
                        if (kBorder) pMin = min(pMin, (colRelevant && ((rowRelevant >> h) & 1u)) ? g : 0x7F800000u);
                        else pMin = min(pMin, g);
                        vPrev[h] = v[h];
                    }
                }
                return pMin;
                };
                auto walkAs = [&](auto sadTag) -> uint32_t {
                    if (border) return restricted ? walk(std::true_type{}, std::true_type{}, sadTag) : walk(std::false_type{}, std::true_type{}, sadTag);
                    return restricted ? walk(std::true_type{}, std::false_type{}, sadTag) : walk(std::false_type{}, std::false_type{}, sadTag);
                };
#ifndef LFG_FOUR_SAD_MAX
#define LFG_FOUR_SAD_MAX 300.0f
#endif
                bool pass;
                if (waveThr < LFG_FOUR_SAD_MAX && !restricted) {       // (noise everywhere; a band's few columns -- the left rim of a pan -- lose by it: pan -0.9 %)
                    const uint32_t sadMin = border ? walk(std::false_type{}, std::true_type{}, std::true_type{}) : walk(std::false_type{}, std::false_type{}, std::true_type{});
                    // (an integer below 2^24: exact as a float; the margin covers the product's rounding and the bound's 10 u)
                    pass = !((float)sadMin > (2.0f * waveThr) * 1.00001f);
                } else {
                    const uint32_t pMin = walkAs(std::false_type{});
                    pass = !(pMin > bitsOf(waveThr * 1.000002f));
                }
                if (restricted) pass = zeroHit(ordL) || pass;          // (the settled pixels outside the band)
                need = need && pass;
#ifndef LFG_FOUR_OFF_AT
#define LFG_FOUR_OFF_AT 48
#endif
                if (fullBatch && __builtin_popcountll(__ballot(need)) >= LFG_FOUR_OFF_AT) useFourPoint = false;
                return need;
        };
#ifndef LFG_SIXTEEN_MAX
#define LFG_SIXTEEN_MAX 2048.0f
#endif
        auto sixteenApplies = [&]() { return useSixteen && !(waveThr < kOnePointMax) && waveThr < LFG_SIXTEEN_MAX; };
        // bit k of the result: candidate i0 + k of the staged order has to be evaluated in full
        // (ordL: this lane's candidate, an entry of the order; need: it has to be looked at; fullBatch: 64 candidates)
        auto latticeBatch = [&](const uint32_t ordL, bool need, const bool fullBatch) -> unsigned long long {
#ifndef LFG_ONEPOINT_OFF
#define LFG_ONEPOINT_OFF 96.0f
#endif
            if (waveThr < LFG_ONEPOINT_OFF) {                          // (a distance is at most 510, but few exceed a threshold of a hundred)
                const lds_ro_u32_ptr w = (lds_ro_u32_ptr)(sWin + kSeg * seg) + (ordL >> 16);
                // the candidate's smallest squared lattice distance, as bits: non-negative floats, whose order is
                // the order of their bit patterns (one v_min_u32 per point, no NaN handling)
                uint32_t dMin = 0x7F800000u;
                auto bitsOf = [](float d) { return __builtin_bit_cast(uint32_t, d); };
                if (border && LFG_ZERO_COMPARE && waveThr < 0.5f) {
                    dMin = zeroHit(ordL) ? 0u : 0x7F800000u;           // (see the interior case below)
                } else if (border) {
#pragma unroll
                    for (int col = 4; col <= kPTW; col += 4) {
                        const uint32_t inImage = (uint32_t)__builtin_amdgcn_readlane((int)valid, col);
#pragma unroll
                        for (int row = 4; row <= 16; row += 4) {
                            // a point outside the image is no point: its distance becomes "at least infinity"
                            // (scalar mask, no branch)
                            const uint32_t drop = (((inImage >> row) & 1u) - 1u) & 0x7F800000u;
                            dMin = min(dMin, bitsOf(distanceOf((uint32_t)__builtin_amdgcn_readlane((int)c[row], col), w[col * kWinH + row])) | drop);
                        }
                    }
                } else {
                    uint32_t tex[kLatCols][kLatRows];    // all 14 texel reads in flight at once: one LDS round trip, not 14
#pragma unroll
                    for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                        for (int t = 0; t < kLatRows; ++t) tex[ci][t] = w[(kLatC0 + 8 * ci) * kWinH + kLatR0 + 8 * t];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (LFG_ZERO_COMPARE && waveThr < 0.5f) {
                        // Every pixel of the wave owns a zero-cost candidate (the threshold words stand for ranks): the test
                        // reads "some lattice distance is exactly 0", and a squared distance is 0 iff the two texels are
                        // the same four bytes -- one compare per point instead of three dot products.  This is the state of
                        // most of a frame under a pan or where nothing moves, seventeen batches per wave.
                        uint32_t acc = 0xFFFFFFFFu;                     // (smallest XOR: see the lookahead below)
#pragma unroll
                        for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                            for (int t = 0; t < kLatRows; ++t)
                                acc = min(acc, tex[ci][t] ^ (uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci));
                        }
                        dMin = acc == 0u ? 0u : 0x7F800000u;
                    } else if (LFG_SAD_TEST && waveThr < LFG_SAD_TEST_MAX) {
                        // (small thresholds: a distance is at least half its SAD -- see the lookahead of the batch loop)
                        uint32_t sad = 0xFFFFFFFFu;
#pragma unroll
                        for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                            for (int t = 0; t < kLatRows; ++t)
                                sad = min(sad, __builtin_amdgcn_sad_u8((uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci), tex[ci][t], 0u));
                        }
                        dMin = (float)sad <= (2.0f * waveThr) * 1.00001f ? 0u : 0x7F800000u;
                    } else {
#pragma unroll
                        for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                            for (int t = 0; t < kLatRows; ++t)
                                dMin = min(dMin, bitsOf(distanceOf((uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci), tex[ci][t])));
                        }
                    }
                }
                // n > waveThr^2 (1 + 2^-21) => sqrt(n) > waveThr (1 + 2^-22), which v_sqrt_f32's 1 ulp cannot bring
                // back to waveThr: every distance S~ would add exceeds the threshold.  (Both factors below round,
                // 2^-24 each, against a margin of 2^-20; a threshold that stands for a zero cost squares to 0 and
                // the test reads n > 0.)
                const float thrSq = (waveThr * waveThr) * 1.000001f;
                need = need & !(dMin > __builtin_bit_cast(uint32_t, thrSq));
            }
            // (only where the one-point test has just let more than an eighth of the batch through, and not any more once
            //  a full batch came out of it three quarters intact: segments that search in full anyway stop paying for it)
            // (where the one-point test does not apply at all -- thresholds of a hundred and more: a match under sensor noise --
            //  every candidate of the batch is still there, and the walk costs as much as evaluating two or three of them in
            //  full: worth it from three on.  The seven candidates that follow the top hint on such frames used to be
            //  evaluated one by one, a sixth of the wave's time.)
#ifndef LFG_FOUR_MIN_WIDE
#define LFG_FOUR_MIN_WIDE 2
#endif
            const int fourFrom = waveThr < LFG_ONEPOINT_OFF ? 8 : LFG_FOUR_MIN_WIDE;
            if (fourPointApplies() && __builtin_popcountll(__ballot(need)) > fourFrom) need = fourPointBatch(ordL, need, fullBatch);
            return __ballot(need);
        };
        // bit k of the result: lane k's candidate (ordL) still has to be evaluated in full
        auto sixteenBatch = [&](const uint32_t ordL, const unsigned long long needMask, const bool fullBatch) -> unsigned long long {
            uint32_t l16;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l16));
            bool need = ((needMask >> l16) & 1ull) != 0ull;
            {
                // SIXTEEN-point test, where four distances do not reach the threshold either: a match that is only nearly
                // exact -- the rows and columns the upscaler filters differently next to the border, compression noise --
                // costs several hundred, a wrong candidate ten thousand, and four of its 64 distances sum to 830 +- 140, of
                // which the smallest of a segment's 75 groups is often below 500.  The block positions with both
                // coordinates even form a lattice of 12 x 32 points of which every pixel's 8 x 8 block holds exactly a
                // 4 x 4 group (rows 2a .. 2a + 6 with a = ceil(r / 2) for pixel row r, columns likewise), and
                // S~ >= (the sum of those sixteen distances) (1 - 10.1 u): S~ adds the same rounded distances in a depth-6
                // tree, the group sum here in four levels.  Mean 3,300: every wrong candidate falls to it while the
                // threshold stays below 2,048.  The lane walks the lattice column by column -- twelve distances, eleven
                // pair sums, nine vertical sums of four, and the running sums over four columns -- 4,400 instructions per
                // batch of 64 candidates: a sixth of evaluating them.  A rolled loop (32 unrolled columns would be 4,000
                // instructions); positions outside the image add nothing, groups that no pixel inside the
                // image maps to are left out.
                // EIGHT points first (round 4): the lattice's every other column -- columns = 0 mod 4, rows even -- still puts
                // 2 x 4 points into every block (columns 4 t0, 4 t0 + 4 with t0 = ceil(i / 4) for pixel column i; rows as above),
                // S~ >= (the sum of those eight distances) (1 - 9.1 u) by the same argument, and the walk is half as long.  Their
                // sum is 1,650 +- 200 for a wrong candidate, of which the smallest of a segment's groups is some 1,100: enough against
                // thresholds below 512 (LFG_EIGHT_MAX) -- the rows above the strip a vertical pan exposes, whose blocks reach into it
                // (386 under the benchmark's pan: 40 -> 21 us per wave in these walks), sensor noise -- and where the threshold
                // is larger, or for the candidates it lets through when they are many, the sixteen-point walk follows as before.
                auto bitsOf = [](float d) { return __builtin_bit_cast(uint32_t, d); };
                int rowsL = ty0 + kSeg * seg;
                asm volatile("" : "+s"(rowsL));
                uint32_t rowRelevant = 0u;                              // bit a: some pixel row inside the image maps to vertical group a
#pragma unroll
                for (int a = 0; a < 9; ++a) rowRelevant |= (!border || rowsL + max(0, 2 * a - 1) < H) ? (1u << a) : 0u;
                const bool restricted = LFG_BAND && (bandLo > 0 || bandHi < kPTW - 1);
                auto walk = [&](auto banded, auto atBorder, auto eightTag) -> uint32_t {     // (eight copies: see the four-point test)
                constexpr bool kBanded = decltype(banded)::value, kBorder = decltype(atBorder)::value, kEight = decltype(eightTag)::value;
                constexpr int kStep = kEight ? 4 : 2, kCols = kEight ? 16 : 32, kClose = kEight ? 1 : 3;   // lattice column stride; columns; a group is closed by its last column
                // groups ceil(i / kStep) of the band's pixel columns i
                const int bLo = (bandLo + kStep - 1) / kStep, bHi = (bandHi + kStep - 1) / kStep;
                lds_ro_u32_ptr w = (lds_ro_u32_ptr)(sWin + kSeg * seg) + (ordL >> 16);
                uint32_t gMin = 0x7F800000u;
                float vOld[9], hOld[2][9];
#pragma unroll
                for (int a = 0; a < 9; ++a) {       // (interior: the groups that the first columns would close are kept out of the minimum by huge sums)
                    vOld[a] = kBorder ? 0.0f : 1.0e30f; hOld[0][a] = kBorder ? 0.0f : 1.0e30f; hOld[1][a] = kBorder ? 0.0f : 1.0e30f;
                }
#pragma unroll 2
                for (int t = 0; t < kCols; ++t, w += kStep * kWinH) {
                    if (kBanded && (t < bLo || t > bHi + kClose)) continue;  // (wave-uniform)
                    int colL = kStep * t;
                    asm volatile("" : "+s"(colL));
                    uint32_t tex[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) tex[k] = w[2 * k];
                    const uint32_t inImage = kBorder ? (uint32_t)__builtin_amdgcn_readlane((int)valid, colL) : 0xFFFFFFFFu;
                    float d[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) {
                        const float dd = __builtin_amdgcn_sqrtf(distanceOf((uint32_t)__builtin_amdgcn_readlane((int)c[2 * k], colL), tex[k]));
                        const uint32_t keep = 0u - ((inImage >> (2 * k)) & 1u);       // scalar: all ones or zero
                        d[k] = kBorder ? __builtin_bit_cast(float, bitsOf(dd) & keep) : dd;
                    }
                    float pr[11], v[9];
#pragma unroll
                    for (int k = 0; k < 11; ++k) pr[k] = d[k] + d[k + 1];
#pragma unroll
                    for (int a = 0; a < 9; ++a) v[a] = pr[a] + pr[a + 2];
                    // sixteen: columns t-1, t; with the pair sum of columns t-3, t-2 the group b = t - 3, whose first pixel column
                    // is max(0, 2 b - 1).  eight: columns t-1, t are the group b = t - 1, first pixel column max(0, 4 b - 3).
                    const bool colRelevant = t >= (kBanded ? bLo : 0) + kClose &&
                                             (!kBorder || tx0 + max(0, kStep * (t - kClose) - (kStep - 1)) < W);
                    const int sel = t & 1;                             // hOld[sel] holds the pair sum made two columns ago
#pragma unroll
                    for (int a = 0; a < 9; ++a) {
                        const float h = vOld[a] + v[a];
                        const uint32_t g = kEight ? bitsOf(h) : bitsOf((sel ? hOld[1][a] : hOld[0][a]) + h);
                        if (kBorder) gMin = min(gMin, (colRelevant && ((rowRelevant >> a) & 1u)) ? g : 0x7F800000u);
                        else gMin = min(gMin, g);
                        if (!kEight) { if (sel) hOld[1][a] = h; else hOld[0][a] = h; }
                        vOld[a] = v[a];
                    }
                }
                return gMin;
                };
                auto walkAs = [&](auto eightTag) -> uint32_t {
                    if (border) return restricted ? walk(std::true_type{}, std::true_type{}, eightTag) : walk(std::false_type{}, std::true_type{}, eightTag);
                    return restricted ? walk(std::true_type{}, std::false_type{}, eightTag) : walk(std::false_type{}, std::false_type{}, eightTag);
                };
#ifndef LFG_EIGHT
#define LFG_EIGHT 1
#endif
#ifndef LFG_EIGHT_MAX
#define LFG_EIGHT_MAX 512.0f            // (1,024: the top and right rim of the pan, thresholds of 620 - 700, let more than a dozen of a batch
                                        //  through and pay for both walks -- pan -1.3 %; 512: pan +1.3 %, noisy +2 %, occluded +1.1 %,
                                        //  moving objects +0.8 %; 400 and 600 within 0.5 % of it)
#endif
#ifndef LFG_SIXTEEN_FROM
#define LFG_SIXTEEN_FROM 12             // (see the batch loop)
#endif
#ifndef LFG_EIGHT_MAX_INSIDE
#define LFG_EIGHT_MAX_INSIDE 1024.0f    // ... away from the border, where such thresholds are heavy noise everywhere (+-8 levels at the input:
                                        //  575 -> 682 frames/s) and not a rim's few rows; a wave whose first full batch leaves more than the
                                        //  sixteen-point walk's worth (LFG_EIGHT_OFF_AT) stops trying
#endif
#ifndef LFG_EIGHT_OFF_AT
#define LFG_EIGHT_OFF_AT 16
#endif
                const bool settledHit = restricted ? zeroHit(ordL) : false;     // (the settled pixels outside the band)
                bool decided = false;
                if (LFG_EIGHT && useEight && waveThr < (border ? LFG_EIGHT_MAX : LFG_EIGHT_MAX_INSIDE)) {
                    const uint32_t eMin = walkAs(std::true_type{});
                    need = need && (!(eMin > bitsOf(waveThr * 1.000002f)) || settledHit);
                    const int left = __builtin_popcountll(__ballot(need));
                    if (fullBatch && left >= LFG_EIGHT_OFF_AT) useEight = false;      // (not decisive here: the sixteen-point walk alone from now on)
                    decided = left <= LFG_SIXTEEN_FROM;                // (few enough to evaluate: the longer walk would cost more)
                }
                if (!decided) {
                    const uint32_t gMin = walkAs(std::false_type{});
                    need = need && (!(gMin > bitsOf(waveThr * 1.000002f)) || settledHit);
                }
                if (fullBatch && __builtin_popcountll(__ballot(need)) >= 48) useSixteen = false;
            }
            return __ballot(need);
        };
        // A window of ONE colour (zero fill included) makes every candidate of every pixel of the segment read the same
        // prev texel at every block position: all 1089 costs are the same operations on the same values, equal to the
        // last bit, and the first candidate in tie order wins whatever curr holds.  Flat areas whose brightness changes
        // (a fade over letterbox bars, a flat background) would otherwise tie at a non-zero cost, overflow the lists
        // and send the tile through the exact kernel.  The answer is carried like a zero-cost one: rank 0 inside the
        // threshold word, no records, and the usual paths turn it into the vector.  (Checked in chunks of eleven
        // rows; textured content leaves after the first.)
        {
            const lds_ro_u32_ptr wb = (lds_ro_u32_ptr)(sWin + kSeg * seg);
            const uint32_t ref = wb[0];
            const int colA = lane, colB = lane + (kWinW - 64);         // 0..63 and 31..94: all 95 window columns
            // (one row first: textured content -- nearly every segment -- leaves after two reads, not twenty-two)
            bool flat = __builtin_amdgcn_readfirstlane(__ballot(wb[colA * kWinH] != ref || wb[colB * kWinH] != ref) == 0ull);
            constexpr int kRows = kSegD + 2 * kR;                      // 55 window rows serve this segment
            for (int r0 = 0; r0 < kRows && flat; r0 += kRows / 5) {
                // (all 22 reads of a chunk in flight, then one OR of XORs: `&&` made each read wait for the one before)
                uint32_t a[kRows / 5], b[kRows / 5];
#pragma unroll
                for (int r = 0; r < kRows / 5; ++r) { a[r] = wb[colA * kWinH + r0 + r]; b[r] = wb[colB * kWinH + r0 + r]; }
                uint32_t diff = 0u;
#pragma unroll
                for (int r = 0; r < kRows / 5; ++r) diff |= (a[r] ^ ref) | (b[r] ^ ref);
                flat = __builtin_amdgcn_readfirstlane(__ballot(diff != 0u) == 0ull);
            }
            if (flat) {
                const float first = __builtin_bit_cast(float, 0x00800000u);          // "zero-cost" word of rank 0
#pragma unroll
                for (int i = 0; i < kRun; ++i) {
                    const float fx = thr2[i].x, fy = thr2[i].y;        // +inf inside the image, -inf outside
                    thr2[i].x = fx > 0.0f ? first : fx;
                    thr2[i].y = fy > 0.0f ? first : fy;
                }
                return 0;
            }
        }
        // ---- Narrow search.  A segment that holds pixels without any match searches in full: no test can drop a
        // candidate for the whole wave, and every candidate costs a full evaluation of all 16 x 56 pixels although only
        // the unmatched ones can still record anything.  Often those lie in a narrow band of columns -- the strip a pan
        // exposes at the left or right edge (ten columns for the benchmark's motion), the vertical edge of a moving
        // object or of an occlusion: two thirds of the full-search waves on the benchmark's pan, a fifth to a third on
        // the occluded and moving-object frames.  Once the hints have been tried the wave looks at its thresholds: if
        // every pixel that is still above LFG_NARROW_THR lies within W <= 16 columns (W a multiple of four), those
        // columns leave the wide machinery -- their thresholds and counts move to LDS, the registers hold -inf, which
        // never passes a test and does not count in waveThr, so the lattice tests work again for the rest of the
        // segment -- and are searched AFTER the wide batches, K = 64 / (W + 8) candidates per pass (5 | 4 | 3 | 2):
        //   * lanes (candidate k, position column j) compute the band's W + 7 position columns for K candidates side
        //     by side: window reads, distances and column sums are the wide code with per-lane offsets;
        //   * the slab holds all sixteen rows (pitch 164, columns interleaved four ways: writes and reads conflict-free);
        //   * lanes (pixel row, group of four columns) then take the K candidates one after the other: eleven reads,
        //     22 adds, four comparisons each, with the pixel's threshold, count and list handled by this one lane as
        //     in the wide code -- nothing shared, no atomics.
        // Distances and column sums, two thirds of an evaluation, are paid once per K candidates, the row sums cover
        // 16 x W pixels instead of 16 x 56.  At the end thresholds and counts go back to the lanes that own the pixels
        // in the wide layout and the epilogue runs unchanged.
        // (After the wide batches, not between them: by then the wide machinery's registers -- the lane's 23
        //  current-frame texels and their squares above all -- are free.  Interleaved, the two sets did not fit and
        //  every pass waited for scratch reloads behind its own record stores.)
        bool narrow = false;
        int nXb = 0, nW = 0;                                           // the band: pixel columns nXb .. nXb + nW - 1 of the tile
        uint32_t *const nThr = sNarrow[wave];                          // [16][kNarrowMax] threshold words (float bits)
        uint32_t *const nCnt = nThr + kSeg * kNarrowMax;               // [16][kNarrowMax] records so far | plateau member seen << 31
        auto entryOf = [&](int idx) { return min(idx < nHead ? idx : eBegin + (idx - nHead), kCand - 1); };
        auto enterNarrow = [&]() {
            int lo = 99, hi = -1;
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const float fx = thr2[i].x, fy = thr2[i].y;            // (-inf: outside the image; +inf: nothing yet)
                if (fx >= LFG_NARROW_THR || fy >= LFG_NARROW_THR) { lo = min(lo, kRun * q + i); hi = max(hi, kRun * q + i); }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
            lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
            if (hi < lo || hi - lo >= kNarrowMax) return;
            nW = ((hi - lo) / 4 + 1) * 4;
            nXb = min(lo, kPTW - nW);
            wave_lds_sync();
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int i = 0; i < kRun; ++i) {
                    const int col = kRun * q + i - nXb;
                    if (col >= 0 && col < nW) {
                        const int cell = (8 * hb + r8) * kNarrowMax + col;
                        const float fx = thr2[i].x, fy = thr2[i].y;
                        nThr[cell] = __builtin_bit_cast(uint32_t, hb ? fy : fx);
                        nCnt[cell] = ((cnt2[hb][i >> 1] >> (16 * (i & 1))) & 0xFFFFu) | (((plateauSeen >> (kRun * hb + i)) & 1u) << 31);
                        if (hb) thr2[i].y = -__builtin_inff(); else thr2[i].x = -__builtin_inff();
                    }
                }
            }
            wave_lds_sync();
            narrow = true;
#ifdef LFG_MOTION_STAMPS
            stampNarrow = (unsigned)(64 / ((nW + kB - 1 + 3) & ~3));
#endif
        };
        auto narrowPhase = [&](int iBegin) -> bool {
            const int iEnd = nEntries;
            const int groups = nW / 4;
            const int colsPer = (nW + kB - 1 + 3) & ~3;                // 12 | 16 | 20 | 24 slab columns per candidate
            const int perPass = 64 / colsPer;                          // 5 | 4 | 3 | 2 candidates
            // compute side: lane = (candidate kc, position column nXb + cl); spare lanes repeat the last one
            int kc = lane / colsPer, cl = lane - kc * colsPer;
            if (kc >= perPass) { kc = perPass - 1; cl = colsPer - 1; }
            const int pcol = min(nXb + cl, kPTW + kB - 2);             // (the padding columns of a candidate compute sums nobody reads)
            const lds_ro_u32_ptr nWin = (lds_ro_u32_ptr)(sWin + pcol * kWinH + kSeg * seg);
            uint32_t cN[kSegD], ccN[kSegD], nValid = 0u;
            {
                const int gx = bx0 + pcol;
                const bool okx = gx >= 0 && gx < W;
                const uint8_t *const column = curr + (size_t)clampi(gx, 0, W - 1) * 4u;
                // (the pitch through an empty asm: sharing the 23 row offsets with the loads at the unit's start keeps 46
                //  scalars alive -- spilled -- from there to here in every unit, narrow or not)
                int pitchN = currPitch;
                asm volatile("" : "+s"(pitchN));
#pragma unroll
                for (int j = 0; j < kSegD; ++j) {
                    const int gy = by0 + kSeg * seg + j;
                    const bool ok = okx && gy >= 0 && gy < H;
                    const uint32_t t = *reinterpret_cast<const uint32_t *>(column + (size_t)clampi(gy, 0, H - 1) * (size_t)pitchN);
                    cN[j] = ok ? t : 0u;
                    nValid |= (ok ? 1u : 0u) << j;
                }
#pragma unroll
                for (int j = 0; j < kSegD; ++j) ccN[j] = __builtin_amdgcn_udot4(cN[j], cN[j], 0x4B000000u, false);
            }
            f32x2 *const nSlabW = reinterpret_cast<f32x2 *>(sSlab[wave]) + ((lane & 3) * kNarrowQ + (lane >> 2));    // slab column = lane
            // read side: lane = (pixel row r, group g of four columns)
            const int r = lane & 15, g = lane >> 4;
            const bool active = g < groups;
            float thrN[4];
            uint32_t cntN[4], platN = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cell = r * kNarrowMax + 4 * (active ? g : 0) + j;
                const uint32_t w = nCnt[cell];
                thrN[j] = active ? __builtin_bit_cast(float, nThr[cell]) : -__builtin_inff();
                cntN[j] = w & 0xFFFFu;
                platN |= (w >> 31) << j;
            }
            // column sum i (0..10) of candidate k for this lane: slab column c = colsPer k + 4 g + i, stored at (c & 3) kNarrowQ + c / 4
            const lds_ro_f32_ptr nSlabR = (lds_ro_f32_ptr)(sSlab[wave] + (r >> 1) * kNarrowPitch + (r & 1) + 2 * (active ? g : 0));
            const uint32_t nLaneOff = (uint32_t)r * (listK * rowStride) + (uint32_t)(nXb + 4 * g);
            const int segY0 = ty0 + kSeg * seg;
            auto testCandidate = [&](const float (&X)[4 + kB - 1], uint32_t ord, uint32_t countIt) {
                float h2[10], h4[8], sN[4];
#pragma unroll
                for (int i = 0; i < 10; ++i) h2[i] = X[i] + X[i + 1];
#pragma unroll
                for (int i = 0; i < 8; ++i) h4[i] = h2[i] + h2[i + 2];
#pragma unroll
                for (int i = 0; i < 4; ++i) sN[i] = h4[i] + h4[i + 4];
                float top = thrN[0] - sN[0];
#pragma unroll
                for (int j = 1; j < 4; ++j) top = __builtin_fmaxf(top, thrN[j] - sN[j]);
                if (__builtin_amdgcn_readfirstlane(__ballot(top >= 0.0f) == 0ull)) return;
                const uint32_t cand = ord & 0xFFFFu;
                const int candDx = (int)((ord >> 16) / (uint32_t)kWinH) - kR, candDy = (int)((ord >> 16) % (uint32_t)kWinH) - kR;
                const uint32_t zeroCap = 0x00800000u + cand;
                const bool candMayLeave = windowLeavesPrev &&
                    block_leaves_prev_any(tx0 + nXb, min(tx0 + nXb + nW - 1, W - 1), segY0, min(segY0 + kSeg - 1, H - 1), candDx, candDy, W, H);
                unsigned long long hit[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) hit[j] = __ballot(sN[j] <= thrN[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (hit[j] != 0ull) {                              // wave-uniform
                        asm volatile("; some lane records a candidate (narrow)");
                        const float s = sN[j];
                        if (s <= thrN[j]) {                            // (the rules of rowSumsAndTest)
                            const float cap = __builtin_fmaxf(s * kRatio, __builtin_bit_cast(float, zeroCap));
                            if (s < thrN[j] * kRestart) cntN[j] = 0u;   // (`restart`: every earlier record is dead)
                            thrN[j] = __builtin_fminf(thrN[j], cap);
                            const uint32_t at = __umul24(min(cntN[j], listK - 1u), rowStride) + nLaneOff + (uint32_t)j;
                            if (s != 0.0f) waveList[at] = rec_make(s, cand);
                            uint32_t inc = (s != 0.0f && countIt != 0u) ? 1u : 0u;
                            if (candMayLeave) {
                                uint32_t l;
                                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l));
                                const bool plateau = block_leaves_prev(tx0 + nXb + 4 * (int)(l >> 4) + j, segY0 + (int)(l & 15u), candDx, candDy, W, H);
                                inc = (plateau && ((platN >> j) & 1u) != 0u) ? 0u : inc;
                                platN |= (plateau && inc != 0u) ? (1u << j) : 0u;
                            }
                            cntN[j] += inc;
                        }
                    }
                }
            };
            uint32_t p[kSegD];
            f32x2 c8[8];
            float v8[kSeg];
            bool pending = false;
            int idx0 = iBegin, idxP = 0, started = 0;
            for (;;) {
                const bool have = idx0 < iEnd && started < 64;
                if (!have && !pending) {                               // (the pipeline has drained) every 64 entries: give up?
                    if (idx0 >= iEnd) break;
                    started = 0;
                    const bool over = cntN[0] >= listK || cntN[1] >= listK || cntN[2] >= listK || cntN[3] >= listK;    // (the last slot is a spare)
                    if (__builtin_amdgcn_readfirstlane(__ballot(over) != 0ull)) sGiveUp = 1u;
                    if (__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&sGiveUp) != 0) return false;
                    continue;
                }
                if (have) {
                    const uint32_t ordC = ((lds_ro_u32_ptr)sOrder)[entryOf(min(idx0 + kc, iEnd - 1))];
                    const lds_ro_u32_ptr w = nWin + (ordC >> 16);
#pragma unroll
                    for (int j = 0; j < kSegD; ++j) p[j] = w[j];
                }
                if (pending) {             // the slab holds the previous pass: sixteen rows of perPass candidates' column sums
                    wave_lds_sync();
#pragma nounroll                   // (one copy of the test, not one per candidate)
                    for (int k = 0; k < perPass && idxP + k < iEnd; ++k) {
                        const lds_ro_f32_ptr base = nSlabR + 2 * (colsPer / 4) * k;
                        float X[4 + kB - 1];
#pragma unroll
                        for (int i = 0; i < 4 + kB - 1; ++i) X[i] = base[2 * ((i & 3) * kNarrowQ + (i >> 2))];
                        const uint32_t ord = ((lds_ro_u32_ptr)sOrder)[entryOf(idxP + k)];
                        testCandidate(X, (uint32_t)__builtin_amdgcn_readfirstlane((int)ord), (idxP + k) >= nHead ? 1u : 0u);
                    }
                    wave_lds_sync();
                }
                __builtin_amdgcn_sched_barrier(0);
                if (have) {
                    columnSums(p, cN, ccN, nValid, c8);
#pragma unroll
                    for (int a = 0; a < 8; ++a) { v8[a] = c8[a].x; v8[a + 8] = c8[a].y; }
                    wave_lds_sync();
#pragma unroll
                    for (int a = 0; a < 8; ++a) nSlabW[a * (kNarrowPitch / 2)] = f32x2{v8[2 * a], v8[2 * a + 1]};
                    wave_lds_sync();
                    started += perPass;
                }
                idxP = idx0; pending = have;
                if (have) idx0 += perPass;
            }
            // the band's final thresholds and counts: through LDS back to the lanes that own the pixels in the wide layout
            wave_lds_sync();
            if (active) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cell = r * kNarrowMax + 4 * g + j;
                    nThr[cell] = __builtin_bit_cast(uint32_t, thrN[j]);
                    nCnt[cell] = min(cntN[j], 0x7FFFu) | (((platN >> j) & 1u) << 31);
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int i = 0; i < kRun; ++i) {
                    const int col = kRun * q + i - nXb;
                    if (col >= 0 && col < nW) {
                        const int cell = (8 * hb + r8) * kNarrowMax + col;
                        const float t = __builtin_bit_cast(float, nThr[cell]);
                        const uint32_t w = nCnt[cell];
                        if (hb) thr2[i].y = t; else thr2[i].x = t;
                        const uint32_t sh = 16u * (uint32_t)(i & 1);
                        cnt2[hb][i >> 1] = (cnt2[hb][i >> 1] & ~(0xFFFFu << sh)) | ((w & 0xFFFFu) << sh);
                        plateauSeen |= (w >> 31) << (kRun * hb + i);
                    }
                }
            }
            return true;
        };
#ifndef LFG_ROW_BAND
#define LFG_ROW_BAND 1
#endif
        // ---- Row band.  The other shape the unmatched pixels of a segment take: a few ROWS over its whole width -- the strip
        // a pan exposes at the top or bottom of the frame (four rows for the benchmark's motion and the three above them, whose
        // blocks reach into it: 134 units of 230 us each, a fifth of a call's workgroup time), the horizontal edge of a moving
        // object or of an occlusion.  Too wide for the narrow search, so every candidate used to cost a full evaluation of all
        // 16 x 56 pixels.  If the pixels still above LFG_NARROW_THR after the hints lie within EIGHT rows, the eight rows
        // rB .. rB + 7 leave the wide machinery as a narrow band's columns do (thresholds parked in LDS, -inf in the registers:
        // the lattice tests work again for the other eight rows) and are searched after the wide batches, TWO candidates per pass:
        //   * lane = position column as in the wide code, but 15 position rows instead of 23, and the two candidates side by
        //     side in the halves of packed registers: one packed subtract / add per position, ONE sliding tree (34 packed adds)
        //     for both, and the eight rows of column sums leave as eight (candidate A, candidate B) pairs per lane;
        //   * every wide lane (r8, q) owns exactly one row of the band -- r8 if r8 >= rB, else r8 + 8 -- so the read side keeps the
        //     wide mapping: thirteen 8-byte reads, ONE packed tree of row sums for both candidates, seven pixels per lane, each
        //     with its threshold, count and list as in the wide code.
        // Same distances, same trees, same S~ to the bit as a full evaluation; two thirds of its instructions per candidate.
        bool rowBand = false;
        int rB = 0;                                                    // the band: pixel rows rB .. rB + 7 of the segment
        auto enterRowBand = [&]() {
            int lo = 99, hi = -1;
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const float fx = thr2[i].x, fy = thr2[i].y;            // (-inf: outside the image; +inf: nothing yet)
                if (fx >= LFG_NARROW_THR) { lo = min(lo, r8); hi = max(hi, r8); }
                if (fy >= LFG_NARROW_THR) { lo = min(lo, r8 + 8); hi = max(hi, r8 + 8); }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); }
            lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
            if (hi < lo || hi - lo >= 8) return;
            rB = min(lo, 8);
            const bool upper = r8 < rB;                                // this lane's row of the band: r8 + 8 (else r8)
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const float fx = thr2[i].x, fy = thr2[i].y;
                nThr[i * 64 + lane] = __builtin_bit_cast(uint32_t, upper ? fy : fx);
                thr2[i].x = upper ? fx : -__builtin_inff();
                thr2[i].y = upper ? -__builtin_inff() : fy;
            }
            wave_lds_sync();
            rowBand = true;
#ifdef LFG_MOTION_STAMPS
            stampNarrow = 7u;
#endif
        };
        static_assert(kRun * 64 <= 2 * kSeg * kNarrowMax, "a row band's thresholds fit a wave's narrow-search words");
        auto rowBandPhase = [&](int iBegin) -> bool {
            constexpr int kBandD = kB + 7;                             // 15 position rows serve eight pixel rows
            const int iEnd = nEntries;
            uint32_t lB;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(lB));
            const int r8B = (int)(lB & 7u), qB = (int)(lB >> 3);
            const bool upper = r8B < rB;
            const int rowInSeg = upper ? r8B + 8 : r8B;                // this lane's pixel row of the segment
            // read side: the lane's seven pixels of that row
            float thrB[kRun];
            uint32_t cntB[kRun], platB;
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                thrB[i] = __builtin_bit_cast(float, nThr[i * 64 + (int)lB]);
                cntB[i] = ((upper ? cnt2[1][i >> 1] : cnt2[0][i >> 1]) >> (16 * (i & 1))) & 0xFFFFu;
            }
            platB = (plateauSeen >> (upper ? kRun : 0)) & ((1u << kRun) - 1u);
            const uint32_t bLaneOff = (uint32_t)rowInSeg * (listK * rowStride) + (uint32_t)(kRun * qB);
            const lds_ro_f32x2_ptr bSlabR = (lds_ro_f32x2_ptr)(sSlab[wave]) + ((r8B - rB) & 7) * (kSlabP / 2) + kRun * qB;
            f32x2 *const bSlabW = reinterpret_cast<f32x2 *>(sSlab[wave]) + (int)lB;
            // compute side: position column `lane`, rows rB .. rB + 14 of the segment's 23
            const int pcol = min((int)lB, kPTW + kB - 2);              // (lane 63 repeats lane 62: sums nobody reads)
            const lds_ro_u32_ptr bWin = (lds_ro_u32_ptr)(sWin + pcol * kWinH + kSeg * seg + rB);
            uint32_t cB[kBandD], ccB[kBandD], validB = 0u;
            {
                const int gx = bx0 + pcol;
                const bool okx = gx >= 0 && gx < W;
                const uint8_t *const column = curr + (size_t)clampi(gx, 0, W - 1) * 4u;
                int pitchB = currPitch;                                // (through an empty asm: see narrowPhase)
                asm volatile("" : "+s"(pitchB));
#pragma unroll
                for (int j = 0; j < kBandD; ++j) {
                    const int gy = by0 + kSeg * seg + rB + j;
                    const bool ok = okx && gy >= 0 && gy < H;
                    const uint32_t t = *reinterpret_cast<const uint32_t *>(column + (size_t)clampi(gy, 0, H - 1) * (size_t)pitchB);
                    cB[j] = ok ? t : 0u;
                    validB |= (ok ? 1u : 0u) << j;
                }
#pragma unroll
                for (int j = 0; j < kBandD; ++j) ccB[j] = __builtin_amdgcn_udot4(cB[j], cB[j], 0x4B000000u, false);
            }
            const int allValid = __builtin_amdgcn_readfirstlane((int)((__ballot(validB != (1u << kBandD) - 1u) & 0x7FFFFFFFFFFFFFFFull) == 0ull));
            const int segY0 = ty0 + kSeg * seg;
            // the eight column sums of two candidates, (A, B) per row: columnSums' arithmetic, the candidates in the register halves
            auto bandSums = [&](const uint32_t (&pa)[kBandD], const uint32_t (&pb)[kBandD], f32x2 (&V)[8]) {
                const f32x2 kBias = {8388608.0f, 8388608.0f};
                f32x2 D[kBandD];
#pragma unroll
                for (int j = 0; j < kBandD; ++j) {
                    const f32x2 F1 = {__builtin_bit_cast(float, __builtin_amdgcn_udot4(pa[j], pa[j], ccB[j], false)),
                                      __builtin_bit_cast(float, __builtin_amdgcn_udot4(pb[j], pb[j], ccB[j], false))};
                    const f32x2 F2 = {__builtin_bit_cast(float, __builtin_amdgcn_udot4(cB[j], pa[j], 0x4B800000u, false)),
                                      __builtin_bit_cast(float, __builtin_amdgcn_udot4(cB[j], pb[j], 0x4B800000u, false))};
                    const f32x2 N = (F1 - F2) + kBias;
                    D[j] = f32x2{__builtin_amdgcn_sqrtf(N.x), __builtin_amdgcn_sqrtf(N.y)};
                }
                if (!allValid) {                  // position outside the image: skipped by the shader, adds 0 here
                    asm volatile("; positions outside the image (row band)");
#pragma unroll
                    for (int j = 0; j < kBandD; ++j) {
                        const bool in = ((validB >> j) & 1u) != 0u;
                        D[j].x = in ? D[j].x : 0.0f; D[j].y = in ? D[j].y : 0.0f;
                    }
                }
                f32x2 B2[kBandD - 1], G4[kBandD - 3];
#pragma unroll
                for (int j = 0; j < kBandD - 1; ++j) B2[j] = D[j] + D[j + 1];
#pragma unroll
                for (int j = 0; j < kBandD - 3; ++j) G4[j] = B2[j] + B2[j + 2];
#pragma unroll
                for (int j = 0; j < 8; ++j) V[j] = G4[j] + G4[j + 4];
            };
            // the rules of rowSumsAndTest for one candidate and the lane's seven pixels
            // (Without a branch per pixel -- every lane doing the bookkeeping by selects, only the record's store predicated --
            //  a bottom-strip unit of the benchmark's pan took 261 us instead of 235: each select waits for a lane mask that went
            //  from the VALU through a scalar AND and back.)
            auto testCandidate = [&](const float (&s)[kRun], uint32_t ord, uint32_t countIt) {
                const uint32_t cand = ord & 0xFFFFu;
                const int candDx = (int)((ord >> 16) / (uint32_t)kWinH) - kR, candDy = (int)((ord >> 16) % (uint32_t)kWinH) - kR;
                const uint32_t zeroCap = 0x00800000u + cand;
                const bool candMayLeave = windowLeavesPrev &&
                    block_leaves_prev_any(tx0, min(tx0 + kPTW - 1, W - 1), segY0 + rB, min(segY0 + rB + 7, H - 1), candDx, candDy, W, H);
                unsigned long long hit[kRun];
#pragma unroll
                for (int i = 0; i < kRun; ++i) hit[i] = __ballot(s[i] <= thrB[i]);
#pragma unroll
                for (int i = 0; i < kRun; ++i) {
                    if (hit[i] != 0ull) {                              // wave-uniform
                        asm volatile("; some lane records a candidate (row band)");
                        const float sv = s[i];
                        if (sv <= thrB[i]) {
#ifdef LFG_DIAG_BAND_NO_RECORDS        // (timing experiment, wrong results: thresholds follow the minima, nothing is recorded)
                            thrB[i] = __builtin_fminf(thrB[i], __builtin_fmaxf(sv * kRatio, __builtin_bit_cast(float, zeroCap)));
#else
                            const float cap = __builtin_fmaxf(sv * kRatio, __builtin_bit_cast(float, zeroCap));
                            if (sv < thrB[i] * kRestart) cntB[i] = 0u;  // (`restart`: every earlier record is dead)
                            thrB[i] = __builtin_fminf(thrB[i], cap);
                            const uint32_t at = __umul24(min(cntB[i], listK - 1u), rowStride) + bLaneOff + (uint32_t)i;
                            if (sv != 0.0f) waveList[at] = rec_make(sv, cand);
                            uint32_t inc = (sv != 0.0f && countIt != 0u) ? 1u : 0u;
                            if (candMayLeave) {
                                uint32_t l;
                                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l));
                                const int rl = (int)(l & 7u);
                                const bool plateau = block_leaves_prev(tx0 + kRun * (int)(l >> 3) + i, segY0 + (rl < rB ? rl + 8 : rl), candDx, candDy, W, H);
                                inc = (plateau && ((platB >> i) & 1u) != 0u) ? 0u : inc;
                                platB |= (plateau && inc != 0u) ? (1u << i) : 0u;
                            }
                            cntB[i] += inc;
#endif
                        }
                    }
                }
            };
            uint32_t pa[kBandD], pb[kBandD];
            bool pending = false;
            int idx0 = iBegin, idxP = 0, started = 0;
            uint32_t ordPA = 0u, ordPB = 0u;
            // (the pass's two entries of the order are fetched a pass ahead, and a pass asks the slab for the previous pass's sums
            //  BEFORE it asks the window for its own texels -- DS operations return in order: fetched where it is needed, each of
            //  the three was a round trip of its own with nothing behind it to hide it)
            auto orderAt = [&](int idx) { return ((lds_ro_u32_ptr)sOrder)[entryOf(min(idx, iEnd - 1))]; };
            uint32_t ordNA = orderAt(idx0), ordNB = orderAt(idx0 + 1);
            for (;;) {
                const bool have = idx0 < iEnd && started < 64;
                if (!have && !pending) {                               // (the pipeline has drained) every 64 entries: give up?
                    if (idx0 >= iEnd) break;
                    started = 0;
                    bool over = false;
#pragma unroll
                    for (int i = 0; i < kRun; ++i) over = over || cntB[i] >= listK;     // (the last slot is a spare)
                    if (__builtin_amdgcn_readfirstlane(__ballot(over) != 0ull)) sGiveUp = 1u;
                    if (__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&sGiveUp) != 0) return false;
                    continue;
                }
                f32x2 X[kRunIn];
                if (pending) {             // the slab holds the previous pass: eight rows of (A, B) column sums
                    wave_lds_sync();
#pragma unroll
                    for (int i = 0; i < kRunIn; ++i) X[i] = bSlabR[i];
                }
                uint32_t ordA = 0u, ordB = 0u;
                if (have) {
                    ordA = (uint32_t)__builtin_amdgcn_readfirstlane((int)ordNA);
                    ordB = (uint32_t)__builtin_amdgcn_readfirstlane((int)ordNB);
                    const lds_ro_u32_ptr wa = bWin + (ordA >> 16), wb = bWin + (ordB >> 16);
#pragma unroll
                    for (int j = 0; j < kBandD; ++j) { pa[j] = wa[j]; pb[j] = wb[j]; }
                    ordNA = orderAt(idx0 + 2); ordNB = orderAt(idx0 + 3);
                }
                if (pending) {
                    f32x2 s2[kRun];
                    runSums(X, s2);
                    float top = -__builtin_inff();
#pragma unroll
                    for (int i = 0; i < kRun; ++i) {
                        const f32x2 dm = f32x2{thrB[i], thrB[i]} - s2[i];
                        top = __builtin_fmaxf(top, __builtin_fmaxf(dm.x, dm.y));
                    }
#ifdef LFG_DIAG_BAND_NO_TEST            // (timing experiment, wrong results: what the evaluation alone costs)
#pragma unroll
                    for (int i = 0; i < kRun; ++i) thrB[i] = __builtin_fminf(thrB[i], __builtin_fminf(s2[i].x, s2[i].y) * kRatio);
                    if (false) {
#else
                    if (!__builtin_amdgcn_readfirstlane(__ballot(top >= 0.0f) == 0ull)) {
#endif
#pragma nounroll                   // (one copy of the test, not one per candidate)
                        for (int k = 0; k < 2 && idxP + k < iEnd; ++k) {
                            float s[kRun];
#pragma unroll
                            for (int i = 0; i < kRun; ++i) s[i] = k ? s2[i].y : s2[i].x;
                            testCandidate(s, k ? ordPB : ordPA, (idxP + k) >= nHead ? 1u : 0u);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (have) {
                    f32x2 V[8];
                    bandSums(pa, pb, V);
                    wave_lds_sync();
#pragma unroll
                    for (int a = 0; a < 8; ++a) bSlabW[a * (kSlabP / 2)] = V[a];
                    wave_lds_sync();
                    started += 2;
                }
                idxP = idx0; ordPA = ordA; ordPB = ordB; pending = have;
                if (have) idx0 += 2;
            }
            // the band's final thresholds and counts: back into the wide layout (each lane's own row)
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const float fx = thr2[i].x, fy = thr2[i].y;
                thr2[i].x = upper ? fx : thrB[i];
                thr2[i].y = upper ? thrB[i] : fy;
                const uint32_t sh = 16u * (uint32_t)(i & 1);
                const uint32_t field = min(cntB[i], 0x7FFFu) << sh, keep = ~(0xFFFFu << sh);
                const uint32_t w0 = cnt2[0][i >> 1], w1 = cnt2[1][i >> 1];
                cnt2[0][i >> 1] = upper ? w0 : ((w0 & keep) | field);
                cnt2[1][i >> 1] = upper ? ((w1 & keep) | field) : w1;
            }
            plateauSeen |= platB << (upper ? kRun : 0);
            return true;
        };
        // Batches: the top hint alone (under a pan its evaluation is all a wave ever evaluates: zero motion, next in the
        // order, then fails the cheap test or is skipped by rank instead of costing a second evaluation); then zero motion
        // and the other hints of this call (at least seven entries, so that units which run the first eight for their
        // thresholds see the same boundary); then sixty-four at a time.
        // (a part of a handed-over segment starts with thresholds that already reflect the hints: full batches from its
        //  first entry on, and the narrow-search decision before the first of them)
        const bool inherited = LFG_QUEUE_INIT && fromQueue;
#ifndef LFG_HINTS_MAX
#define LFG_HINTS_MAX (2 + 62)
#endif
        const int hintsEnd = inherited ? 0 : nHead ? kHead : max(kHead, min((int)orderHints, LFG_HINTS_MAX));
        int firstBatchSurvivors = 0;
        // BY RANK.  Once every pixel of the wave owns a zero-cost candidate only candidates that come EARLIER in the tie
        // order than the latest of those can still matter (zeroBound).  Walking on through the visiting order would
        // look at every remaining batch for the few lanes whose rank qualifies; instead the wave turns to the ranks
        // themselves -- 0 .. zeroBound - 1, sixty-four per batch, through the inverse of the order (sInv), leaving out
        // what it has visited already or what belongs to another part of the order.  Under a pan or on a static frame
        // that is zeroBound / 64 batches instead of seventeen; the candidates looked at are the same.
        bool byRank = false;
        int rank0 = 0, visited = 0;
        // DEFERRED sixteen-point test.  Where a match is only nearly exact everywhere -- sensor or compression noise: real
        // video -- the four-point test leaves a dozen candidates of every batch, and running the sixteen-point test for
        // them costs its 4,400 instructions with four fifths of the lanes idle: nineteen batches of both tests were 90 %
        // of such a wave's time.  So the four-point survivors of a batch wait (128 entries of LDS per wave) until 64 of
        // them are together, and the sixteen-point test runs with every lane busy: a fifth as often.  (A test may run at
        // any later time: thresholds only fall.  Up to the first full batch the survivors are tested at once -- the
        // hand-over decision counts that batch's.)
        uint32_t *const pend = sPending[wave];
        int pendCount = 0;
        if (LFG_QUEUE_INIT && fromQueue) {                             // the thresholds of the wave that handed the segment over
            const uint32_t *const init = sp.dynInit + (size_t)(unit / (sp.dynParts / 4)) * (size_t)(kSeg * kPTW);
            uint32_t theirs[2][kRun];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int i = 0; i < kRun; ++i)
                    theirs[hb][i] = __hip_atomic_load(init + (2 * i + hb) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int i = 0; i < kRun; ++i) {                            // (pixels outside the image keep -inf)
                const float fx = thr2[i].x, fy = thr2[i].y;
                if (fx > 0.0f) thr2[i].x = __builtin_bit_cast(float, theirs[0][i]);
                if (fy > 0.0f) thr2[i].y = __builtin_bit_cast(float, theirs[1][i]);
            }
            refreshZeroBound();
        }
        const bool mayHandOver = whole && !fromQueue && orderHandOver != 0u;
#ifdef LFG_STAMP_LATTICE3
        unsigned long long stampT[4] = {0ull, 0ull, 0ull, 0ull};
        stampT[0] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef LFG_STAMP_LATTICE4
        stampU[1] = __builtin_amdgcn_s_memrealtime();
#endif
        for (int i0 = 0, count = inherited ? 64 : LFG_FIRST_BATCH;;) {
            const bool flushOnly = byRank ? rank0 >= (int)min(zeroBound, (uint32_t)kCand) : i0 >= nEntries;
            if (flushOnly && pendCount == 0) break;
            // (a wave with the whole order turns to the ranks as soon as it can, even where that saves no batch: consecutive
            //  ranks are neighbouring window offsets, so the lattice reads of a batch fall into 64 different LDS banks, while
            //  64 consecutive entries of the visiting order hit the same bank three or four times)
            // (Round 4 also sent whole-order waves whose thresholds are small but NOT zero-cost words -- 58 % of the interior waves
            //  under the benchmark's pan -- to the ranks, all 1089 of them: conflict-free reads against the pseudo-random order's
            //  three to a bank.  Measured 1.3 % slower, 3,100 against 3,145 frames/s: dropped.)
            if (!flushOnly && !byRank && i0 >= max(nHead, LFG_FIRST_BATCH) && zeroBound < (uint32_t)kCand &&
                ((whole && LFG_RANK_ALWAYS) || (int)zeroBound + 64 <= nEntries - i0)) {
                byRank = true; rank0 = 0; visited = i0;
            }
            // (the two batches at which a wave decides how to go on: hand-over, narrow search)
            const bool decisionPoint = !flushOnly && !byRank && (i0 == hintsEnd || i0 == hintsEnd + 64);
            // Hand the segment over?  Either no threshold to test against once every hint has been tried, or the test
            // let a quarter of the first full batch through.
            if (decisionPoint && mayHandOver &&
                ((i0 == hintsEnd && !(waveThr < 4.0f * 510.0f)) || (i0 == hintsEnd + 64 && firstBatchSurvivors >= 16))) {
                // LFG_DYN_PARTS parts of the candidate order, four per queue entry (one workgroup each); the entries of a
                // segment are consecutive slots, so its private lists are the blocks 4 slot .. 4 slot + parts - 1
                const uint32_t kEntries = (uint32_t)sp.dynParts / 4u;
                uint32_t slot = 0u;
                if (lane == 0) slot = atomicAdd(sp.queueCount, kEntries);
                slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
                if (slot + kEntries <= (uint32_t)sp.queueCap) {
                    if (LFG_QUEUE_INIT) {
                        // what this wave knows after the hints is where the segment's parts start: each of them would
                        // otherwise evaluate the head of the order for its thresholds, eight evaluations in each of eight
                        // waves.  (Agent scope: the parts run on other XCDs.  Every word is some candidate's evaluated cost
                        // x kRatio, or a zero-cost word: a valid bound for its pixel in any part.)
                        // They have to be in place before the entries that announce the segment.  Exchanges, not stores plus a
                        // fence: an agent-scope atomic has been performed where every XCD sees it by the time it returns, and the
                        // wave-wide test on the returned values below cannot run before all of them have -- whereas
                        // __threadfence() writes back and invalidates this XCD's whole L2, once per handed-over segment (650 times
                        // a launch on the moving-object frames), and every workgroup staging a window there pays for it.
                        uint32_t *const init = sp.dynInit + (size_t)(slot / kEntries) * (size_t)(kSeg * kPTW);
                        uint32_t returned = 0u;
#pragma unroll
                        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                            for (int i = 0; i < kRun; ++i) {
                                const float fx = thr2[i].x, fy = thr2[i].y;
#ifdef LFG_DIAG_PUSH_FENCE             // (timing experiment: round 2's stores and fence)
                                __hip_atomic_store(init + (2 * i + hb) * 64 + lane, __builtin_bit_cast(uint32_t, hb ? fy : fx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                                returned |= atomicExch(init + (2 * i + hb) * 64 + lane, __builtin_bit_cast(uint32_t, hb ? fy : fx));     // (lane-major: whole lines)
#endif
                            }
                        }
#ifdef LFG_DIAG_PUSH_FENCE
                        __threadfence();
#endif
                        asm volatile("" : "+v"(returned));
                        if (__ballot(returned == 0x7FC12345u) == ~0ull) return 2;      // (never: a NaN pattern no threshold holds; what counts is that every lane's exchanges have returned)
                    }
                    if (lane == 0) {
                        sp.openList[atomicAdd(sp.openCount, 1u)] = (uint32_t)(tile * (kPTH / kSeg) + seg);                 // (its parts leave it to the resolve kernel)
                        sp.segMap[tile * (kPTH / kSeg) + seg] = (4u * slot) | ((uint32_t)sp.dynParts << 24) | (1u << 31);    // (read by the resolve kernel)
                        // the entry itself is the "slot filled" signal (never 0): an atomic, like the read that waits for it
                        for (uint32_t k = 0; k < kEntries; ++k)
                            atomicExch(&sp.queue[slot + k], (uint32_t)tile | ((4u * k) << 20) | ((uint32_t)sp.dynParts << 24) | (1u << 28) | ((uint32_t)seg << 29));
                    }
                    return 2;
                }
            }
            if (LFG_NARROW && decisionPoint && !narrow && i0 == hintsEnd && !(waveThr < LFG_NARROW_THR)) {
                enterNarrow();
                if (LFG_ROW_BAND && !narrow) enterRowBand();
                if (narrow || rowBand) refreshZeroBound();    // the largest threshold of the pixels that stay wide
            }
            if (LFG_BAND && decisionPoint && i0 == hintsEnd && !(waveThr < kOnePointMax)) computeBand();
#ifdef LFG_MOTION_STAMPS
            if (!byRank && i0 == hintsEnd && !narrow) {
                int lo = 99, hi = -1, rlo = 99, rhi = -1;
#pragma unroll
                for (int i = 0; i < kRun; ++i) {
                    const float fx = thr2[i].x, fy = thr2[i].y;
                    if (fx >= 2048.0f) { lo = min(lo, q); hi = max(hi, q); rlo = min(rlo, r8); rhi = max(rhi, r8); }
                    if (fy >= 2048.0f) { lo = min(lo, q); hi = max(hi, q); rlo = min(rlo, r8 + 8); rhi = max(rhi, r8 + 8); }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_xor(lo, off)); hi = max(hi, __shfl_xor(hi, off)); rlo = min(rlo, __shfl_xor(rlo, off)); rhi = max(rhi, __shfl_xor(rhi, off)); }
                stampBox = (unsigned)(lo & 0xFF) | ((unsigned)(hi & 0xFF) << 8) | ((unsigned)(rlo & 0xFF) << 16) | ((unsigned)(rhi & 0xFF) << 24);
                if (hi < lo && stampBand) stampBox = (stampBand & 0xFFFFu) | (0xEEu << 16);      // (no such pixel: the band of the unsettled ones instead)
            }
            if (i0 == hintsEnd) stampThr = __builtin_bit_cast(uint32_t, waveThr);
            stampThrEnd = __builtin_bit_cast(uint32_t, waveThr);
            stampFour = useFourPoint ? 1u : 0u;
#endif
            bool firstFull = false, flushNow = false;
            uint32_t ordL = 0u;            // this lane's candidate: an entry of the order; bit 31: count it when it is recorded
            unsigned long long m = 0ull;   // the lanes whose candidate has to be evaluated in full
            uint32_t l;                    // lane number; volatile, so that it is not hoisted out of the loop and spilled
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l));
            if (!flushOnly) {
                // this lane's candidate `ahead` entries (ranks) beyond the batch's first
                auto candidateAt = [&](int ahead, uint32_t &ord) -> bool {
                    if (!byRank) {
                        const int idx = i0 + ahead + (int)l;
                        ord = ((lds_ro_u32_ptr)sOrder)[min(idx < nHead ? idx : eBegin + (idx - nHead), kCand - 1)];
                        const bool nd = (ahead != 0 || (int)l < count) && idx < nEntries && (ord & 0xFFFFu) < zeroBound;
                        ord |= idx >= nHead ? 0x80000000u : 0u;
                        return nd;
                    }
                    const int r = rank0 + ahead + (int)l;
                    const int rc = min(r, kCand - 1);
                    const int e = (int)sInv[rc];                                      // where the order visits rank r
                    if (rankIsScan) {
                        // rank = scan index = 33 (dy + R) + (dx + R): the entry the order holds for it, without asking the order -- the
                        // window reads then wait for nothing but this arithmetic, and `e` only decides whether the lane counts
                        // (three dependent LDS round trips per pass were two too many)
                        const uint32_t dyi = ((uint32_t)rc * 1986u) >> 16;            // rc / 33 for rc < 1089
                        const uint32_t dxi = (uint32_t)rc - 33u * dyi;
                        ord = (uint32_t)rc | ((dxi * (uint32_t)kWinH + dyi) << 16) | 0x80000000u;
                    } else {
                        ord = ((lds_ro_u32_ptr)sOrder)[e] | 0x80000000u;
                    }
                    return r < (int)zeroBound && e >= eBegin && e < eEnd && nHead + (e - eBegin) >= visited;
                };
                // LOOKAHEAD.  While the thresholds are small (the one-point test alone decides, and nearly always "no") a
                // batch is three LDS round trips of latency around a few dozen instructions -- and a workgroup's time on
                // its CU slot, not its instruction count, is what a frame costs once frames overlap.  So
                // the wave first asks about the next 192 candidates at once -- three per lane (LFG_LOOKAHEAD), their window reads in flight
                // together, the current-frame texels of the lattice points fetched once for both -- and if none of them
                // has to be looked at, skips both batches.  Otherwise the batches are taken one by one as usual.
                constexpr int kAhead = LFG_LOOKAHEAD;                  // candidates per lane (0: off)
#ifdef LFG_STAMP_LATTICE
                const unsigned long long ta0 = __builtin_amdgcn_s_memrealtime();
                if (!stampAheadFirst && kAhead > 1 && !border && waveThr < kOnePointMax && pendCount < 64 && (byRank || (count == 64 && i0 != hintsEnd))) stampAheadFirst = ta0;
#endif
                // (From the second batch on, whatever its length: once the thresholds are this small the decisions taken at the
                //  end of the hints -- hand-over, narrow search, the band -- cannot apply any more, thresholds only fall.)
                if (kAhead > 1 && !border && waveThr < kOnePointMax && pendCount < 64 && (byRank || i0 >= LFG_FIRST_BATCH) &&
                    (byRank ? rank0 + 64 * (kAhead - 1) < (int)min(zeroBound, (uint32_t)kCand) : i0 + count + 64 * (kAhead - 2) < nEntries)) {
                    constexpr int kA = kAhead > 1 ? kAhead : 1;
                    uint32_t ordA[kA];
                    bool needA[kA];
                    uint32_t tex[kA][kLatCols][kLatRows];
#pragma unroll
                    for (int a = 0; a < kA; ++a) needA[a] = candidateAt(a == 0 ? 0 : (byRank ? 64 : count) + 64 * (a - 1), ordA[a]);
#pragma unroll
                    for (int a = 0; a < kA; ++a) {
                        const lds_ro_u32_ptr w = (lds_ro_u32_ptr)(sWin + kSeg * seg) + ((ordA[a] & 0x7FFFFFFFu) >> 16);
#pragma unroll
                        for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                            for (int t = 0; t < kLatRows; ++t) tex[a][ci][t] = w[(kLatC0 + 8 * ci) * kWinH + kLatR0 + 8 * t];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    bool keep[kA];                     // some lattice distance does not exceed the wave's largest threshold
                    if (waveThr < 0.5f) {              // (zero-cost thresholds: "exceeds" means "differs"; see latticeBatch)
                        // ("some point is the same four bytes" as the smallest XOR being 0: v_xor + half a v_min3 per point and
                        //  nothing but VALU.  A compare per point puts a lane mask into VCC that an s_or has to pick up before
                        //  the next compare may write it -- 48 VALU -> SALU -> VALU hand-overs per pass, four times the cycles.)
                        uint32_t accA[kA];
#pragma unroll
                        for (int a = 0; a < kA; ++a) accA[a] = 0xFFFFFFFFu;
#pragma unroll
                        for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                            for (int t = 0; t < kLatRows; ++t) {
                                const uint32_t cT = (uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci);
#pragma unroll
                                for (int a = 0; a < kA; ++a) accA[a] = min(accA[a], tex[a][ci][t] ^ cT);
                            }
                        }
#pragma unroll
                        for (int a = 0; a < kA; ++a) keep[a] = accA[a] == 0u;
                    } else if (LFG_SAD_TEST && waveThr < LFG_SAD_TEST_MAX) {
                        // Small but not zero thresholds -- a match up to a rounding of the upscaler, a level here and there:
                        // more than half of the interior waves of the benchmark's pan -- need no distance either: a distance
                        // is at least half the sum of its four absolute differences (Cauchy-Schwarz), so "some lattice point
                        // with SAD <= 2 thr" is a necessary condition too, one v_sad_u8 and half a v_min3 per point instead
                        // of three dot products and two adds (wrong candidates have SADs in the hundreds).
                        uint32_t sadA[kA];
#pragma unroll
                        for (int a = 0; a < kA; ++a) sadA[a] = 0xFFFFFFFFu;
#pragma unroll
                        for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                            for (int t = 0; t < kLatRows; ++t) {
                                const uint32_t cT = (uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci);
#pragma unroll
                                for (int a = 0; a < kA; ++a) sadA[a] = min(sadA[a], __builtin_amdgcn_sad_u8(cT, tex[a][ci][t], 0u));
                            }
                        }
                        const float sadMax = (2.0f * waveThr) * 1.00001f;
#pragma unroll
                        for (int a = 0; a < kA; ++a) keep[a] = (float)sadA[a] <= sadMax;
                    } else {
                        uint32_t dMinA[kA];
#pragma unroll
                        for (int a = 0; a < kA; ++a) dMinA[a] = 0x7F800000u;
#pragma unroll
                        for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                            for (int t = 0; t < kLatRows; ++t) {
                                const uint32_t cT = (uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci);
#pragma unroll
                                for (int a = 0; a < kA; ++a) dMinA[a] = min(dMinA[a], __builtin_bit_cast(uint32_t, distanceOf(cT, tex[a][ci][t])));
                            }
                        }
                        const uint32_t thrSqBits = __builtin_bit_cast(uint32_t, (waveThr * waveThr) * 1.000001f);
#pragma unroll
                        for (int a = 0; a < kA; ++a) keep[a] = !(dMinA[a] > thrSqBits);
                    }
                    bool any = false;
#pragma unroll
                    for (int a = 0; a < kA; ++a) any = any | (needA[a] && keep[a]);
                    if (__ballot(any) == 0ull) {
#ifdef LFG_MOTION_STAMPS
                        stampBatches += (unsigned)kA;
#endif
                        if (byRank) rank0 += 64 * kA; else { i0 += count + 64 * (kA - 1); count = 64; }
#ifdef LFG_STAMP_LATTICE
                        stampAhead += __builtin_amdgcn_s_memrealtime() - ta0;
                        stampPass += 1u;
#endif
                        continue;
                    }
#ifdef LFG_STAMP_LATTICE
                    stampPass += 0x10000u;
#endif
                    if (false) {
                    }
                }
                bool need = candidateAt(0, ordL);
#ifdef LFG_STAMP_LATTICE4
                if (!byRank && i0 == 0) { asm volatile("" :: "v"(ordL)); stampU[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
#ifdef LFG_STAMP_LATTICE     // (experiment: o[5] carries the time spent in the lattice tests instead of the first batch's end)
                const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef LFG_STAMP_PHASES
                const unsigned long long tp0 = __builtin_amdgcn_s_memrealtime();
#endif
                m = latticeBatch(ordL & 0x7FFFFFFFu, need, byRank || count == 64);
#ifdef LFG_STAMP_PHASES
                phaseLattice += __builtin_amdgcn_s_memrealtime() - tp0;
#endif
#ifdef LFG_STAMP_LATTICE
                stampLattice += __builtin_amdgcn_s_memrealtime() - tl0;
#endif
                firstFull = !byRank && i0 == hintsEnd;
                if (m == 0ull && pendCount < 64) {     // the common case: the tests dropped the whole batch and nothing is due
                    if (firstFull) firstBatchSurvivors = 0;
#ifdef LFG_MOTION_STAMPS
                    stampBatches += 1u;
                    if (!byRank && i0 == 0) stampFirst = __builtin_amdgcn_s_memrealtime();
#endif
                    if (byRank) rank0 += 64;
                    else { i0 += count; count = i0 == LFG_FIRST_BATCH ? hintsEnd - LFG_FIRST_BATCH : 64; }
                    continue;
                }
#ifndef LFG_DEFER_FROM
#define LFG_DEFER_FROM 0                // a batch's survivors wait in the list when they are more than this many (0: always -- a
                                        // lone survivor evaluated on the spot is an evaluation without anything to overlap with: window
                                        // reads, column sums, slab round trip and row sums one after the other; a dozen together at
                                        // the end of the search run software-pipelined.  8 until round 3: noisy frames +3 % with
                                        // frames in flight, +8 % one at a time)
#endif
#ifndef LFG_SIXTEEN_FROM
#define LFG_SIXTEEN_FROM 12             // ... and the sixteen-point test runs on more than this many of them (its 4,400 instructions
                                        // are a dozen evaluations)
#endif
                if (sixteenApplies() && __builtin_popcountll(m) > LFG_DEFER_FROM) {     // the survivors wait for company
                    const uint32_t at = (uint32_t)pendCount + (uint32_t)__builtin_popcountll(m & ((1ull << l) - 1ull));
                    wave_lds_sync();
                    if ((m >> l) & 1ull) pend[at] = ordL;
                    wave_lds_sync();
                    pendCount += __builtin_popcountll(m);
                    m = 0ull;
                    flushNow = !byRank && i0 <= hintsEnd;                  // (up to the first full batch: tested at once)
                }
                if (firstFull && !flushNow) firstBatchSurvivors = __builtin_popcountll(m);
#ifdef LFG_MOTION_STAMPS
                stampBatches += 1u;
#endif
            }
            // two rounds of full evaluations: this batch's survivors, then -- when 64 candidates wait, or at the end -- the
            // first 64 of the list after their sixteen-point test
#ifdef LFG_STAMP_LATTICE3
            if (!byRank && i0 == 0) stampT[1] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef LFG_STAMP_LATTICE4
            if (!byRank && i0 == 0) { stampU[3] = __builtin_amdgcn_s_memrealtime();
                stampPass7 = ((stampU[0] - stampStaged) & 0xFFFFull) | (((stampU[1] - stampStaged) & 0xFFFFull) << 16) | (((stampU[2] - stampStaged) & 0xFFFFull) << 32) | (((stampU[3] - stampStaged) & 0xFFFFull) << 48); }
#endif
            for (int round = 0; round < 2; ++round) {
                if (round == 1) {
                    if (!(pendCount >= 64 || flushNow || (flushOnly && pendCount > 0))) break;
                    const int take = min(pendCount, 64);
                    wave_lds_sync();
                    ordL = pend[min((int)l, take - 1)];
                    const uint32_t moved = pend[min((int)l + 64, 127)];
                    wave_lds_sync();
                    if ((int)l + 64 < pendCount) pend[l] = moved;      // the rest moves up
                    wave_lds_sync();
                    pendCount -= take;
                    m = __ballot((int)l < take && (ordL & 0xFFFFu) < zeroBound);
#ifdef LFG_STAMP_PHASES
                    const unsigned long long tp1 = __builtin_amdgcn_s_memrealtime();
#endif
                    if (sixteenApplies() && __builtin_popcountll(m) > LFG_SIXTEEN_FROM) m = sixteenBatch(ordL & 0x7FFFFFFFu, m, take == 64);
#ifdef LFG_STAMP_PHASES
                    phaseSixteen += __builtin_amdgcn_s_memrealtime() - tp1;
#endif
                    if (flushNow && firstFull) firstBatchSurvivors = __builtin_popcountll(m);
                    flushNow = false;
                }
#ifdef LFG_MOTION_STAMPS
                stampEvals += (unsigned)__builtin_popcountll(m);
#endif
                if (m == 0ull) continue;
#ifdef LFG_STAMP_PHASES
                const unsigned long long tp2 = __builtin_amdgcn_s_memrealtime();
#endif
                // The survivors, software-pipelined: the window reads of one are in flight while the previous one is
                // finished; a last pass drains the pipeline.
                bool pending = false;
                uint32_t ordP = 0u, cntP = 0u;
                while (m != 0ull || pending) {
                    const bool have = m != 0ull;
                    uint32_t ord = 0u, cntIt = 0u;
                    if (have) {
                        const int b = __builtin_ctzll(m);
                        m &= m - 1ull;
                        ord = (uint32_t)__builtin_amdgcn_readlane((int)ordL, b);
                        cntIt = ord >> 31;
                        ord &= 0x7FFFFFFFu;
                        fetchWindow(p, ord);
                    }
                    if (pending) rowSumsAndTest(x, ordP, cntP);
                    __builtin_amdgcn_sched_barrier(0);
                    bool evaluate = have;
                    if (LFG_EXACT_MATCH && have && !byRank && i0 < hintsEnd) {
                        // EXACT MATCH.  A hint that IS the motion of this segment -- a pan's top hint, zero motion where nothing
                        // moves, an object's own vector -- reads the very bytes of the current frame at every block position:
                        // all 23 x 63 distances are exactly 0, so S~ = 0 for each of the 16 x 56 pixels and the evaluation
                        // could only find that out the long way (46 dot products, 23 square roots, two trees, two slab round
                        // trips).  23 XORs and a ballot do: then every pixel takes the candidate as a zero-cost one -- the
                        // update of rowSumsAndTest for s == 0: rank into the threshold word, no record, no count.  (Positions
                        // outside the image add nothing to S~ and are left out here; lane 63 owns no position column.)  Tried
                        // for the call's hints only: elsewhere it would cost every evaluation 7 % for nothing.
                        uint32_t diff = 0u;
#pragma unroll
                        for (int j = 0; j < kSegD; ++j) diff |= p[j] ^ c[j];
                        if (border) {
                            diff = 0u;
#pragma unroll
                            for (int j = 0; j < kSegD; ++j) diff |= (p[j] ^ c[j]) & (0u - ((valid >> j) & 1u));
                        }
                        if (__builtin_amdgcn_readfirstlane((int)((__ballot(diff != 0u) & 0x7FFFFFFFFFFFFFFFull) == 0ull))) {
                            const float zc = __builtin_bit_cast(float, 0x00800000u + (ord & 0xFFFFu));
#pragma unroll
                            for (int i = 0; i < kRun; ++i) {       // (-inf, a pixel outside the image, stays)
                                thr2[i].x = __builtin_fminf(thr2[i].x, zc);
                                thr2[i].y = __builtin_fminf(thr2[i].y, zc);
                            }
                            evaluate = false;
#ifdef LFG_MOTION_STAMPS
                            stampEvals += 0x10000u;
#endif
                        }
                    }
                    if (evaluate) {
                        columnSums(p, c, cc, valid, v8);
                        transpose(v8, x, slabR);
                    }
                    ordP = ord; cntP = cntIt; pending = evaluate;
                }
#ifdef LFG_STAMP_PHASES
                phaseEval += __builtin_amdgcn_s_memrealtime() - tp2;
#endif
                // lists full somewhere in the tile: stop early
                if (__builtin_amdgcn_readfirstlane(__ballot(listsOverflowed()) != 0ull)) sGiveUp = 1u;
                if (__builtin_amdgcn_readfirstlane((int)*(volatile uint32_t *)&sGiveUp) != 0) return 1;
#ifdef LFG_STAMP_LATTICE3
                if (!byRank && i0 == 0 && round == 0) stampT[2] = __builtin_amdgcn_s_memrealtime();
#endif
                refreshZeroBound();
#ifdef LFG_STAMP_LATTICE3
                if (!byRank && i0 == 0 && round == 0) { stampT[3] = __builtin_amdgcn_s_memrealtime();
                    stampPass7 = ((stampT[0] - stampStaged) & 0xFFFFull) | (((stampT[1] - stampStaged) & 0xFFFFull) << 16) | (((stampT[2] - stampStaged) & 0xFFFFull) << 32) | (((stampT[3] - stampStaged) & 0xFFFFull) << 48); }
#endif
            }
            if (flushOnly) continue;           // (until the list is empty)
#ifdef LFG_MOTION_STAMPS
            if (!byRank && i0 == 0) stampFirst = __builtin_amdgcn_s_memrealtime();
#endif
            if (byRank) rank0 += 64;
            else { i0 += count; count = i0 == LFG_FIRST_BATCH ? hintsEnd - LFG_FIRST_BATCH : 64; }
        }
        if (narrow && !narrowPhase(hintsEnd)) return 1;
        if (rowBand && !rowBandPhase(hintsEnd)) return 1;
        if (!narrow && !rowBand) settledAtZero = waveThr < 0.5f;       // (every pixel of the wave owns a zero-cost candidate)
        return __builtin_amdgcn_readfirstlane(__ballot(listsOverflowed()) != 0ull) ? 1 : 0;
    };
    const int outcome = run();
    const bool gaveUp = outcome == 1;
#ifdef LFG_MOTION_STAMPS
    const unsigned long long stampRunEnd = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef LFG_MOTION_STAMPS
    if (lane == 0 && (fromQueue ? sp.units + unit : unit) < 8192) {
        unsigned long long *o = gMotionStamps + ((size_t)(fromQueue ? sp.units + unit : unit) * 4 + wave) * 8;
        o[0] = stampStart; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = stampEvals; o[3] = ((unsigned long long)borderTile << 32) | stampBatches | (((stampRunEnd - stampStart) & 0x3FFFFFFFull) << 33);
#ifdef LFG_STAMP_LATTICE
        stampStaged = stampStart + stampAhead;        // ("staging": the time in lookahead passes that skipped; "first batch": in the lattice tests)
        stampFirst = stampStaged + stampLattice;
#ifdef LFG_STAMP_LATTICE2
        stampStaged = stampAheadFirst ? stampAheadFirst : stampStart;     // ("staging": when the first lookahead pass began)
#endif
#endif
        o[4] = stampStaged; o[5] = stampFirst; o[6] = ((unsigned long long)stampBox << 32) | stampNarrow | (stampFour << 8) | ((unsigned long long)(fromQueue ? 1u : 0u) << 9) | ((unsigned long long)seg << 10) | ((unsigned long long)tileX << 12) | ((unsigned long long)tileY << 20);
        o[7] = ((unsigned long long)stampThrEnd << 32) | stampThr;
#ifdef LFG_STAMP_LATTICE
        o[7] = stampPass;
#endif
#if defined(LFG_STAMP_LATTICE3) || defined(LFG_STAMP_LATTICE4)
        o[7] = stampPass7;
#endif
#ifdef LFG_STAMP_PHASES
        o[5] = (phaseLattice & 0xFFFFFull) | ((phaseSixteen & 0xFFFFFull) << 20) | ((phaseEval & 0xFFFFFull) << 40);
#endif
    }
#endif
    if (gaveUp) {
        // Flags live on the exact kernel's 64 x 64 tile grid (cleared before this launch): this tile spans
        // one or two of its columns.  Racing writers all store 1.  Sibling waves that already finished have
        // written their thresholds and counts; the resolve kernel ignores flagged tiles.
        if (lane == 0) {
            sGiveUp = 1u;
            const int ex0 = tx0 / kTW, ex1 = min(tx0 + kPTW - 1, W - 1) / kTW;
            // (the number of flagged tiles decides how the exact kernel shares them out; it follows the queue length;
            // the first kShareBelow of them are also listed, right behind the count)
            for (int ex = ex0; ex <= ex1; ++ex) {
                if (atomicExch(&tileFlags[tileY * flagTilesX + ex], 1u) == 0u) {
                    const uint32_t slot = atomicAdd(sp.queueCount + 1, 1u);
                    if (slot < (uint32_t)kShareBelow) sp.queueCount[2 + slot] = (uint32_t)(tileY * flagTilesX + ex);
                }
            }
        }
        if (!segUnit) return;              // (the waves of a segment unit meet at a barrier below)
    }
    if (outcome == 2) return;              // handed over (whole tiles only: no barrier below for them)
    // Threshold and count of every pixel, for the resolve kernel -- which never looks at a segment that settles all of
    // its pixels below, so such a segment (most of a frame under a pan) does not write them at all.
    auto writeThresholds = [&]() {
        // wave-uniform bases (whole tiles: the image-shaped arrays; shared tiles: the unit's private block) plus 32-bit lane
        // offsets, like the lists: a per-lane 64-bit pointer here was spilled, and its reload -- s_waitcnt vmcnt(0) --
        // made each of the 28 stores wait for the one before (20 us per wave that writes its thresholds)
        const size_t first = whole ? (size_t)(ty0 + kSeg * seg) * (size_t)W + (size_t)tx0
                                   : ((size_t)auxUnit * auxRows + (size_t)(kSeg * seg - auxRow0)) * (size_t)kPTW;
        float *const uBase = (whole ? uminOut : auxUminBase) + first;
        uint32_t *const cBase = (whole ? countOut : auxCountBase) + first;
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int py = ty0 + kSeg * seg + 8 * hb + r8;
            const uint32_t off = (uint32_t)(8 * hb + r8) * rowStride + (uint32_t)(kRun * q);
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                if (!gaveUp && py < H && px0 + i < W) {
                    const uint32_t cnt = (cnt2[hb][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                    uBase[off + (uint32_t)i] = hb ? thr2[i].y : thr2[i].x;
                    cBase[off + (uint32_t)i] = cnt;
                }
            }
        }
    };
    const bool settlesHere = whole && !windowLeavesPrev;
    bool thresholdsNeeded = !settlesHere;  // (one call further down: the fourteen stores per lane are inlined once)
    // Whole tiles away from the rim settle their easy pixels here and now, while threshold and count are still in
    // registers: a zero-cost candidate (encoded in the threshold), a single record, or two records of which one
    // survives ARE the answer (motion_resolve_kernel's rules; no plateaus where the window stays inside prev).
    // A segment whose pixels were all settled says so, and the resolve kernel skips the blocks it covers.
    if (whole && !windowLeavesPrev) {
        // the first two records of the pixels that have any (a zero-cost pixel has none: most of a frame under a pan)
        const bool fastSettle = __builtin_amdgcn_readfirstlane((int)(allInside && settledAtZero)) != 0;
        Rec ra[2][kRun], rb[2][kRun];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < kRun; ++i) { ra[hb][i] = 0u; rb[hb][i] = 0u; }
        }
        if (!fastSettle) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const uint32_t cnt = (cnt2[hb][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                if (cnt >= 1u) ra[hb][i] = waveList[laneOff[hb] + (uint32_t)i];
                if (cnt >= 2u) rb[hb][i] = waveList[rowStride + laneOff[hb] + (uint32_t)i];
            }
        }
        }
        uint32_t best[2][kRun];
        bool allSettled = true;
        if (fastSettle) {
            // the common case under a pan or where nothing moves: every threshold word IS the answer (rank + 0x00800000);
            // no records to look at, no pixel outside the image -- a dozen instructions instead of 250
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const float fx = thr2[i].x, fy = thr2[i].y;            // (named floats: see refreshZeroBound)
                best[0][i] = __builtin_bit_cast(uint32_t, fx) - 0x00800000u;
                best[1][i] = __builtin_bit_cast(uint32_t, fy) - 0x00800000u;
            }
        } else
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int py = ty0 + kSeg * seg + 8 * hb + r8;
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const float fx = thr2[i].x, fy = thr2[i].y;            // (named floats: see refreshZeroBound)
                const float bound = hb ? fy : fx;
                const uint32_t cnt = (cnt2[hb][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                const bool inImage = py < H && px0 + i < W;
                const bool sa = rec_cost_low(ra[hb][i]) <= bound, sb = rec_cost_low(rb[hb][i]) <= bound;
                uint32_t b = 0xFFFFFFFFu;
                if (bound < 0.5f) b = __builtin_bit_cast(uint32_t, bound) - 0x00800000u;
                else if (cnt == 1u) b = rec_cand(ra[hb][i]);
                else if (cnt == 2u && sa != sb) b = sa ? rec_cand(ra[hb][i]) : rec_cand(rb[hb][i]);
                best[hb][i] = inImage ? b : 0xFFFFFFFFu;
                allSettled = allSettled && (!inImage || b != 0xFFFFFFFFu);
            }
        }
        // Only a segment that settles ALL of its pixels writes vectors (the resolve kernel redoes every pixel of any
        // other segment anyway), and it writes them as rows: the 16 x 56 vectors go through the wave's slab and leave
        // as 112 contiguous bytes per image row, two rows per store instruction -- a lane's own 2 x 7 pixels, stored
        // from where they were computed, touched sixteen cache lines per instruction with two bytes each (133 MB of
        // write requests per 4K frame for 16.6 MB of vectors).
        if (__builtin_amdgcn_readfirstlane(__ballot(!allSettled) == 0ull)) {
            uint16_t *const rows = reinterpret_cast<uint16_t *>(sSlab[wave]);          // [16][56] vectors (1,792 of 2,176 bytes)
            // (the lane number goes through an empty asm: the slab addresses derived from it are the same for every unit,
            //  and hoisted out of the workgroup's loop over units they are spilled and reloaded here behind a wait)
            int laneE = lane;
            asm volatile("" : "+v"(laneE));
            const int r8E = laneE & 7, qE = laneE >> 3;
            wave_lds_sync();
            // (all fourteen table reads first: interleaved with the slab stores -- through a generic pointer, which the
            //  compiler must assume may alias -- each read waited for the one before, fourteen L2 round trips in a row)
            int scanOf[2][kRun];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int i = 0; i < kRun; ++i)
                    scanOf[hb][i] = (int)rank2scan[best[hb][i] != 0xFFFFFFFFu ? best[hb][i] : 0u];     // (outside the image: never stored)
            }
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
                for (int i = 0; i < kRun; ++i) {
                    const int scan = scanOf[hb][i];
                    const int dyi = scan / kSide, dxi = scan - dyi * kSide;
                    rows[(8 * hb + r8E) * kPTW + kRun * qE + i] = (uint16_t)(uint8_t)(int8_t)(dxi - kR) | (uint16_t)((uint16_t)(uint8_t)(int8_t)(dyi - kR) << 8);
                }
            }
            wave_lds_sync();
            const bool wide = ((mvPitch & 3) == 0) && (((uintptr_t)mv & 3u) == 0u);       // 4-byte stores need 4-byte rows
            const int half = laneE / 28, d = laneE - half * 28;                          // 28 dwords = one row of the tile
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int row = 2 * rr + half, py = ty0 + kSeg * seg + row, px = tx0 + 2 * d;
#ifdef LFG_DIAG_NO_MV_STORE            // (timing experiment, wrong results: what writing the settled segments' vectors costs)
                if (false) {
#else
                if (laneE < 56 && py < H && px < W) {
#endif
                    const uint32_t two = *reinterpret_cast<const uint32_t *>(rows + row * kPTW + 2 * d);
                    if (kFused && sp.fused.data) {           // (wave-uniform) the north-star order: the generated pixels of these two vectors
                        fused_pixel(sp.fused, prev, prevPitch, curr, currPitch, W, H, px, py, (int)(int8_t)(two & 0xFFu), (int)(int8_t)((two >> 8) & 0xFFu));
                        if (px + 1 < W) fused_pixel(sp.fused, prev, prevPitch, curr, currPitch, W, H, px + 1, py, (int)(int8_t)((two >> 16) & 0xFFu), (int)(int8_t)(two >> 24));
                    }
                    int8_t *dst = mv + (size_t)py * (size_t)mvPitch + (size_t)px * 2u;
                    if (kFused && !sp.fused.storeMv) {
                    } else if (wide && px + 1 < W) *reinterpret_cast<uint32_t *>(dst) = two;
                    else {
                        *reinterpret_cast<uint16_t *>(dst) = (uint16_t)two;
                        if (px + 1 < W) *reinterpret_cast<uint16_t *>(dst + 2) = (uint16_t)(two >> 16);
                    }
                }
            }
            if (lane == 0) segDone[tile * (kPTH / kSeg) + seg] = 1u;
        } else {
            thresholdsNeeded = true;
        }
    }
#ifndef LFG_LATE_THRESHOLDS
#define LFG_LATE_THRESHOLDS 1            // a segment unit that pools its four parts writes thresholds and counts only if a pixel stays open
#endif
    // (a segment unit whose four waves hold all parts of the order learns below whether any pixel is left to the resolve
    //  kernel; if none is -- most rim segments under a pan -- its 4 x 7 KB of thresholds and counts are never read: they were
    //  25 MB of the prefilter's 53 MB of writes per 4K frame, against 17 MB of vectors)
    const bool pooledUnit = LFG_LATE_THRESHOLDS && segUnit && nChunks == 4;
    if (thresholdsNeeded && !pooledUnit) writeThresholds();
    // A segment that is left to the resolve kernel goes onto its list, once: by the wave that owns it (whole tiles), by the
    // unit that holds the first part of the order (tiles shared between units), or -- segment units -- further down.
    if (thresholdsNeeded && !gaveUp && !segUnit && chunk == 0 && lane == 0)
        sp.openList[atomicAdd(sp.openCount, 1u)] = (uint32_t)(tile * (kPTH / kSeg) + seg);
    // (eight parts, two workgroups: no pooled settling below.  Plan units only: a handed-over segment was listed by the wave
    //  that pushed it -- listing it again here resolved it twice and could run past the list's tiles * 4 words)
    if (segUnit && nChunks != 4 && !fromQueue && !gaveUp && chunk == 0 && lane == 0)
        sp.openList[atomicAdd(sp.openCount, 1u)] = (uint32_t)(tile * (kPTH / kSeg) + seg);
    // A segment unit does the same for its segment, its four waves pooling what each learnt about its part of the
    // candidate order: the tightest threshold of the four is the pixel's bound, every wave holds its own records
    // against it, and a pixel with exactly one survivor among all four (or a zero-cost candidate) is settled --
    // unless the survivor stands for a plateau, which the resolve kernel sorts out.  The exchange goes through the
    // window's LDS, which every wave has finished with at the first barrier.
    if (segUnit && nChunks == 4) {         // (all parts of the order in this workgroup)
        constexpr int kPix = 2 * kRun * 64;                            // pixel slots of a wave (896)
        float *const sBound = reinterpret_cast<float *>(sWin);        // [4][kPix] thresholds
        uint32_t *const sMine = sWin + 4 * kPix;                       // [4][kPix] 0: no survivor, 0x80000000 | rank: one, ~0: unknown
        uint32_t *const sOpen = sWin + 8 * kPix;                       // some pixel of the segment is left to the resolve kernel
        static_assert(8 * kPix + 1 <= kWinH * kWinW, "exchange area fits the window");
        __syncthreads();
        if (*(volatile uint32_t *)&sGiveUp != 0u) return;              // the tile goes through the exact kernel
        Rec ra[2][kRun], rb[2][kRun];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const uint32_t cnt = (cnt2[hb][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                ra[hb][i] = 0u; rb[hb][i] = 0u;
                if (cnt >= 1u) ra[hb][i] = waveList[laneOff[hb] + (uint32_t)i];
                if (cnt >= 2u) rb[hb][i] = waveList[rowStride + laneOff[hb] + (uint32_t)i];
                const float fx = thr2[i].x, fy = thr2[i].y;            // (named floats: see refreshZeroBound)
                sBound[wave * kPix + (kRun * hb + i) * 64 + lane] = hb ? fy : fx;
            }
        }
        if (tid == 0) *sOpen = 0u;
        __syncthreads();
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const int at = (kRun * hb + i) * 64 + lane;
                const float bound = __builtin_fminf(__builtin_fminf(sBound[at], sBound[kPix + at]),
                                                    __builtin_fminf(sBound[2 * kPix + at], sBound[3 * kPix + at]));
                const uint32_t cnt = (cnt2[hb][i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
                const bool sa = cnt >= 1u && rec_cost_low(ra[hb][i]) <= bound;
                const bool sb = cnt >= 2u && rec_cost_low(rb[hb][i]) <= bound;
                uint32_t word = 0u;
                if (cnt > 2u || (sa && sb)) word = 0xFFFFFFFFu;
                else if (sa) word = 0x80000000u | rec_cand(ra[hb][i]);
                else if (sb) word = 0x80000000u | rec_cand(rb[hb][i]);
                sMine[wave * kPix + at] = word;
            }
        }
        __syncthreads();
        // (three passes over this wave's share of the fourteen slots -- decide, read the table, store -- so that the table
        //  reads are in flight together: read and stored slot by slot, each read waited for the store before it)
        bool open = false;
        uint32_t bestOf[2][kRun];
        bool singleOf[2][kRun];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int py = ty0 + kSeg * seg + 8 * hb + r8;
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                bestOf[hb][i] = 0xFFFFFFFFu; singleOf[hb][i] = false;
                if (((kRun * hb + i) & 3) != wave) continue;           // the fourteen slots are dealt out to the four waves
                if (!(py < H && px0 + i < W)) { bestOf[hb][i] = 0xFFFFFFFEu; continue; }      // (outside the image: nothing to do)
                const int at = (kRun * hb + i) * 64 + lane;
                const float bound = __builtin_fminf(__builtin_fminf(sBound[at], sBound[kPix + at]),
                                                    __builtin_fminf(sBound[2 * kPix + at], sBound[3 * kPix + at]));
                if (bound < 0.5f) {
                    bestOf[hb][i] = __builtin_bit_cast(uint32_t, bound) - 0x00800000u;
                } else {
                    uint32_t n = 0u, any = 0u;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t word = sMine[w * kPix + at];
                        n += word != 0u ? 1u : 0u;
                        any |= word;
                    }
                    // exactly one wave reports, and it reports a single rank
                    if (n == 1u && any != 0xFFFFFFFFu) { bestOf[hb][i] = any & 0x7FFFFFFFu; singleOf[hb][i] = true; }
                }
            }
        }
        int scanOfSeg[2][kRun];
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                scanOfSeg[hb][i] = 0;
                if (((kRun * hb + i) & 3) != wave) continue;
                scanOfSeg[hb][i] = (int)rank2scan[bestOf[hb][i] < (uint32_t)kCand ? bestOf[hb][i] : 0u];
            }
        }
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const int py = ty0 + kSeg * seg + 8 * hb + r8;
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                if (((kRun * hb + i) & 3) != wave) continue;
                if (bestOf[hb][i] == 0xFFFFFFFEu) continue;            // outside the image
                bool settled = bestOf[hb][i] != 0xFFFFFFFFu;
                if (settled) {
                    const int scan = scanOfSeg[hb][i];
                    const int dyi = scan / kSide, dxi = scan - dyi * kSide;
                    if (singleOf[hb][i] && block_leaves_prev(px0 + i, py, dxi - kR, dyi - kR, W, H)) settled = false;
                    if (settled) {
                        if (kFused && sp.fused.data) fused_pixel(sp.fused, prev, prevPitch, curr, currPitch, W, H, px0 + i, py, dxi - kR, dyi - kR);
                        int8_t *dst = mv + (size_t)py * (size_t)mvPitch + (size_t)(px0 + i) * 2u;
                        if (!kFused || sp.fused.storeMv)
                        *reinterpret_cast<uint16_t *>(dst) = (uint16_t)(uint8_t)(int8_t)(dxi - kR) | (uint16_t)((uint16_t)(uint8_t)(int8_t)(dyi - kR) << 8);   // both components, one store
                    }
                }
                open = open || !settled;
            }
        }
        if (__builtin_amdgcn_readfirstlane(__ballot(open) != 0ull) && lane == 0) *sOpen = 1u;
        __syncthreads();
        if (tid == 0) {
            if (*sOpen == 0u) segDone[tile * (kPTH / kSeg) + seg] = 1u;
            else sp.openList[atomicAdd(sp.openCount, 1u)] = (uint32_t)(tile * (kPTH / kSeg) + seg);
        }
        if (pooledUnit && *(volatile uint32_t *)sOpen != 0u) writeThresholds();
    }
#ifdef LFG_MOTION_STAMPS
    if (lane == 0 && (fromQueue ? sp.units + unit : unit) < 8192)      // (the end of the unit with its epilogue)
        gMotionStamps[((size_t)(fromQueue ? sp.units + unit : unit) * 4 + wave) * 8 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
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
constexpr uint32_t kNoUnit = 0xFFFFFFFFu;
#ifndef LFG_QUEUE_FIRST
#define LFG_QUEUE_FIRST 0
#endif
#ifndef LFG_PREF_POLL_SLEEPS
#define LFG_PREF_POLL_SLEEPS 2         // x 8128 clocks between two looks of a waiting workgroup
#endif
// A word that workgroups on other XCDs update with atomics, as they left it: an agent-scope load (served by L2, which
// those atomics write through), and every 32nd look a read-modify-write that changes nothing -- a compare-and-swap
// against a value the word never holds -- so that progress never depends on a cache line being refreshed.
__device__ __forceinline__ uint32_t peek(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t peek_hard(uint32_t *p) { return atomicCAS(p, 0xFFFFFFFFu, 0xFFFFFFFFu); }

template <bool kFused>
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
            if (LFG_QUEUE_FIRST && next == kNoUnit && owned == kNoUnit && peek(&ctrl[2]) < min(peek(sp.queueCount), cap)) {
                const uint32_t h = atomicAdd(&ctrl[2], 1u);
                if (h < min(peek_hard(sp.queueCount), cap)) { entry = wait_entry(h); next = 0x80000000u | h; }
                else if (h < cap) owned = h;              // lost the race for the last filled slot: h is mine when it fills
            }
            if (next == kNoUnit) {        // (drawn when needed, not ahead: a workgroup holding two of the long units in a row
                                          //  would run them one after the other while others idle)
                const uint32_t u = atomicAdd(&ctrl[0], 1u);
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
                if (owned == kNoUnit && peek_hard(sp.queueCount) != 0u) { const uint32_t h = atomicAdd(&ctrl[2], 1u); if (h < cap) owned = h; }
                uint32_t n = 0u;
                while (owned != kNoUnit) {
                    const bool hard = (++n & 31u) == 0u;
                    if ((entry = hard ? peek_hard(&sp.queue[owned]) : peek(&sp.queue[owned])) != 0u) { next = 0x80000000u | owned; owned = kNoUnit; break; }
                    if ((hard ? peek_hard(&ctrl[1]) : peek(&ctrl[1])) >= planUnits) {
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
        prefilter_unit<kFused>(prev, prevPitch, curr, currPitch, W, H, list, uminOut, countOut, tileFlags, flagTilesX, order32, sp,
                       mv, mvPitch, rank2scan, segDone, (int)(next & 0x7FFFFFFFu), fromQueue, entry, sWin, sSlab, sOrder, sInv, sGiveUp, sNarrow, sPending);
        if (!fromQueue) {
            __syncthreads();               // every wave of the unit is past its pushes
            if (threadIdx.x == 0) atomicAdd(&ctrl[1], 1u);
        }
    }
}

// Exact chain of motion.comp:27-47 for ONE candidate of ONE pixel (literal loops, same distance
// arithmetic as everywhere else in this file).
__device__ float exact_cost(const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr,
                            int currPitch, int W, int H, int px, int py, int dx, int dy) {
    float diff = 0.0f;
    for (int y = 0; y < kB; ++y) {
        const int cy = py - kB / 2 + y;
        if (cy < 0 || cy >= H) continue;
        for (int x = 0; x < kB; ++x) {
            const int cx = px - kB / 2 + x;
            if (cx < 0 || cx >= W) continue;
            const uint32_t c = *reinterpret_cast<const uint32_t *>(curr + (size_t)cy * (size_t)currPitch + (size_t)cx * 4u);
            const int qx = cx + dx, qy = cy + dy;
            uint32_t p = 0u;
            if (qx >= 0 && qy >= 0 && qx < W && qy < H)
                p = *reinterpret_cast<const uint32_t *>(prev + (size_t)qy * (size_t)prevPitch + (size_t)qx * 4u);
            const float cc[4] = {unorm8_to_float(byte0(c)), unorm8_to_float(byte1(c)),
                                 unorm8_to_float(byte2(c)), unorm8_to_float(byte3(c))};
            const f32x4 pp = {unorm8_to_float(byte0(p)), unorm8_to_float(byte1(p)),
                              unorm8_to_float(byte2(p)), unorm8_to_float(byte3(p))};
            diff += dist4<true>(cc, pp);
        }
    }
    return diff;
}

// The resolve kernel's launch: a fixed number of waves that share out the rows of the segments the prefilter left open -- the
// compact list its units wrote as they ended (sp.openList, sp.openCount): item i = row i % 16 of open segment i / 16, lane =
// pixel column of the tile (56 of 64 lanes).  Round 2 launched one workgroup
// per 64 x 4 pixels of the frame -- 32,400 at 4K -- of which nine in ten read three flags and left: 51 us under a pan for ~550
// segments with work.  Nothing waits for anything here: the list is complete when this launch starts.
constexpr int kResolveGroups = 1024;
template <bool kFused>
__global__ __launch_bounds__(256) void motion_resolve_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H, const Rec *__restrict__ list,
    const float *__restrict__ uminIn, const uint32_t *__restrict__ countIn, const uint32_t *__restrict__ tileFlags,
    int tilesX, PrefilterPlan sp, const uint32_t *__restrict__ rank2scan, const uint32_t *__restrict__ segDone) {
    __shared__ float sDist[4][kB * kB];    // one block of distances per wave (cooperative exact evaluation)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // How the rows get to the waves -- wave w takes items w, w + waves, ... -- is decided by the GRID (launch_motion_prefiltered_8_16):
    //   * a context that runs one frame at a time launches kResolveGroups workgroups, and every row is dealt out: nothing but
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
#ifdef LFG_MOTION_STAMPS
    const unsigned long long stampT0 = __builtin_amdgcn_s_memrealtime();
#endif
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
    const bool live = inside && flagged == 0u && !settledSeg;
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
#ifdef LFG_MOTION_STAMPS
    const unsigned long long stampT1 = __builtin_amdgcn_s_memrealtime();
    const int stampTodo = __builtin_popcountll(todo);
    if (survivors > 1u) { atomicAdd(&gResolveStats[0], 1ull); atomicAdd(&gResolveStats[1], (unsigned long long)survivors);
                          atomicAdd(&gResolveStats[2], (unsigned long long)(px < 64 || py < 64 || px >= W - 64 || py >= H - 64)); }
#endif
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
#ifdef LFG_MOTION_STAMPS
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        atomicAdd(&gResolveStats[4], 1ull);                     // working waves
        atomicAdd(&gResolveStats[5], stampT1 - stampT0);        // gather time (10 ns ticks)
        atomicAdd(&gResolveStats[6], t2 - stampT1);             // cooperative + plateau + write time
        atomicMax(&gResolveStats[7], t2 - stampT0);             // longest wave
        atomicAdd(&gResolveStats[8], (unsigned long long)stampTodo);
        atomicMin(&gResolveStats[9], stampT0); atomicMax(&gResolveStats[10], t2);
    }
#endif
    }
    }   // items
}

// tileFlags == nullptr: the literal kernel for every tile.  Otherwise the second pass of the prefiltered path: one launch,
// whatever was flagged (see the kernel).
hipError_t launch_motion_tiled_8_16(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                    const lfg_frame &mv, const uint32_t *tileFlags, const uint32_t *rank2scan,
                                    unsigned long long *merge, uint32_t *flaggedTiles, const FusedOut &fused,
                                    bool expectNothing, uint32_t *verdictWord, uint32_t *hostWord) {
    const int tilesX = ((int)curr.width + kTW - 1) / kTW, tilesY = ((int)curr.height + kTH - 1) / kTH;
    // (expectNothing: the lane's previous call flagged no tile -- 64 workgroups, which take it all if this one does after all)
    const dim3 grid = tileFlags ? dim3(expectNothing ? 64u : (unsigned)std::max(tilesX * tilesY, kShareBelow * kFallbackParts), 1, 1)
                                : dim3((unsigned)tilesX, (unsigned)tilesY, 1);
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, grid, dim3(kNT), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height, tileFlags, rank2scan,
                           merge, flaggedTiles, tilesX, tilesX * tilesY, fused, verdictWord, hostWord);
    };
    const bool loops = tileFlags != nullptr && expectNothing;
    if (fused.data) { if (loops) launch(motion_tiled_8_16_loop_kernel<true>); else launch(motion_tiled_8_16_kernel<true>); }
    else            { if (loops) launch(motion_tiled_8_16_loop_kernel<false>); else launch(motion_tiled_8_16_kernel<false>); }
    return hipGetLastError();
}

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
    l.order = align(l.queue + (size_t)l.queueCap * sizeof(uint32_t));           // this call's hints and visiting order
    // work-unit tables and the auxiliary arrays of the shared tiles (see prefilter_plan): one 56 x 64 block per unit
    // (two plans side by side where the lean kernel may run -- frames in flight: the one for calls that go through it, rimSplit2,
    //  and the one for calls that do not; the host picks per call, lfg_capi.cpp: motion_run.  The auxiliary arrays serve either.)
    const PrefilterPlanHost plan = prefilter_plan(width, height, slots, rimSplit);
    const PrefilterPlanHost plan2 = rimSplit2 ? prefilter_plan(width, height, slots, rimSplit2) : PrefilterPlanHost();
    const size_t auxUnits = (size_t)std::max(plan.auxUnits, plan2.auxUnits);
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
#ifndef LFG_LEAN_XCD_BANDS
#define LFG_LEAN_XCD_BANDS 1             // workgroup i of that launch lands on XCD i mod 8: give every XCD a contiguous band of tiles, in raster
#endif                                   // order, so that the windows of neighbouring tiles (they overlap 2.7 x) meet in ONE L2
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
#ifndef LFG_LEAN_PARTIAL
#define LFG_LEAN_PARTIAL 1
#endif
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

int prefilter_slots() {
    int dev = 0, cus = 0, perCu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, motion_prefilter_kernel<false>, kPNT, 0) != hipSuccess) return 0;
    return cus * perCu;
}

// Tables of the 8/16 motion paths for one tie-break rule (kMotionTableWords uint32 each, see lfg_internal.hpp):
//   rank2scan[r], r = 0..kCand   scan index (dy+R)*33 + (dx+R) of the candidate with rank r in the tie order;
//                                reference semantics: r itself (motion.comp's scan, first strict minimum wins);
//                                intended semantics: candidates sorted by dx*dx + dy*dy, then scan order, so equal
//                                costs resolve to the shortest vector.  Entry kCand is the sentinel kCand.
//   order32[e], e = 0..kCand-1   the prefilter's visiting order: rank in the low half, offset of the candidate in
//                                the LDS window, (dx+R)*kWinH + (dy+R), in the high half.
void motion_tables(bool intended, uint32_t *rank2scan, uint32_t *order32, uint32_t *entryOfScan, uint32_t *baseScan) {
    uint16_t byRank[kCand], rankOf[kCand];
    for (int i = 0; i < kCand; ++i) byRank[i] = (uint16_t)i;
    if (intended) {
        auto d2 = [](int scan) { const int dy = scan / kSide - kR, dx = scan % kSide - kR; return dx * dx + dy * dy; };
        std::stable_sort(byRank, byRank + kCand, [&](uint16_t a, uint16_t b) { return d2(a) < d2(b); });
    }
    for (int r = 0; r < kCand; ++r) { rank2scan[r] = byRank[r]; rankOf[byRank[r]] = (uint16_t)r; }
    rank2scan[kCand] = kCand;
    // A fixed pseudo-random permutation of the scan indices (Fisher-Yates driven by a 32-bit LCG).  Visiting the
    // candidates in an order unrelated to their position makes the sequence of costs behave like a random sample, so
    // a pixel sees only ~ln(1089) running minima -- for smooth content and for image borders alike, where a
    // spatially ordered walk would keep finding slightly better candidates and overflow the lists.
    uint16_t order[kCand];
    for (int i = 0; i < kCand; ++i) order[i] = (uint16_t)i;
    uint32_t state = 0x9E3779B9u;
    for (int i = kCand - 1; i > 0; --i) {
        state = state * 1664525u + 1013904223u;
        const int j = (int)((state >> 8) % (uint32_t)(i + 1));
        const uint16_t t = order[i]; order[i] = order[j]; order[j] = t;
    }
    // ... except that zero motion goes first: static areas (where m = 0 costs exactly 0) then close their
    // threshold at once, before the exactly tied candidates at the rim of a flat area can fill the lists.
    for (int i = 0; i < kCand; ++i) {
        if (order[i] == kR * kSide + kR) { order[i] = order[0]; order[0] = (uint16_t)(kR * kSide + kR); break; }
    }
    for (int scan = 0; scan < kCand; ++scan) {
        const uint32_t dyi = scan / kSide, dxi = scan % kSide;
        entryOfScan[scan] = rankOf[scan] | ((dxi * kWinH + dyi) << 16);
    }
    for (int i = 0; i < kCand; ++i) {
        baseScan[i] = order[i];
        order32[i] = entryOfScan[order[i]];
    }
}

// ------------------------------------------------------------------------------ per-call visiting order
//
// The fewer running minima a pixel sees, the less the prefilter records.  Two tiny kernels put the candidates
// that are likely to be the answer at the front of the visiting order of THIS call: motion_hint_kernel block-matches
// 256 sample blocks (a 16 x 16 grid over the frame) with a plain SAD over all 1089 candidates and reports each
// block's best; motion_order_kernel emits the most popular hint, zero motion, the other distinct hints, then the
// rest of the fixed pseudo-random order.  With a few dominant motions in the frame (a pan, a handful of objects) most pixels meet
// their minimum within the first few candidates and close their thresholds at once.  The order only changes what
// is recorded along the way: every candidate is still evaluated for every pixel and the results are identical.

constexpr int kHintWin = kB + 2 * kR;                // 40 x 40 texels of prev around a sample block

// 256 threads, four or five candidates each, at most 128 VGPRs: a workgroup fits into the room ONE workgroup of a running
// prefilter launch leaves on a CU.  With 1,024 threads (one or two candidates each, 10 us on an idle chip instead of 13)
// it needed a CU with no prefilter workgroup at all, and with frames in flight (lanes) the next frame's hints waited
// 300-500 us for one (kernel trace; the frame rate is the same -- the prefilter's workgroup slots are the bound -- but the
// lanes' latency is not).  A context that runs one frame at a time has the chip to itself: there the kernel is launched
// with 1,024 threads (round 3: a call under a pan 0.328 -> 0.324 ms).
#ifndef LFG_HINT_THREADS
#define LFG_HINT_THREADS 256
#endif
constexpr int kHintThreadsInFlight = LFG_HINT_THREADS;
#ifndef LFG_HINT_THREADS_ALONE
#define LFG_HINT_THREADS_ALONE 1024      // a context that runs one frame at a time: nothing else wants the CU, and a call's latency is what counts
#endif
constexpr int kHintThreadsAlone = LFG_HINT_THREADS_ALONE;

// One launch does two jobs (round 2: a memset in front of it): every workgroup first clears its share of the call's control
// area (tile flags, segment marks and map, counters, queue: `clearWords` words from `clearFrom`) -- the prefilter launch
// behind this one is what reads them -- then block-matches one sample block.  The ordering of the 256 hints stays a launch of
// its own (motion_order_kernel, one workgroup): folded into the workgroup that finishes last here it cost the occluded frames
// 5 - 15 % with three frames in flight (983 -> 905 frames/s with its tables in LDS, which no longer fits beside two resident
// prefilter workgroups; 837 with a 1.2 KB version that does) for 5 us of one call's latency.
template <int kHintThreads>
__global__ __launch_bounds__(kHintThreads, kHintThreads <= 256 ? 4 : 1) void motion_hint_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int W, int H, uint32_t *__restrict__ hints, uint32_t *__restrict__ clearFrom, int clearWords) {
    __shared__ uint32_t sP[kHintWin * kHintWin];
    __shared__ uint32_t sC[kB * kB], sValid[kB * kB];
    __shared__ uint32_t sBest;
    const int tid = threadIdx.x;
    for (int i = (int)blockIdx.x * kHintThreads + tid; i < clearWords; i += (int)gridDim.x * kHintThreads) clearFrom[i] = 0u;
    const int gx = blockIdx.x % kHintGrid, gy = blockIdx.x / kHintGrid;
    const int bx = (2 * gx + 1) * W / (2 * kHintGrid) - kB / 2, by = (2 * gy + 1) * H / (2 * kHintGrid) - kB / 2;
    if (tid == 0) sBest = 0xFFFFFFFFu;
    {   // both loads of a thread in flight at once (branch-free: clamped address, value dropped outside the image)
        constexpr int kRounds = (kHintWin * kHintWin + kHintThreads - 1) / kHintThreads;
        uint32_t v[kRounds];
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int i = min(k * kHintThreads + tid, kHintWin * kHintWin - 1);
            const int x = bx - kR + i % kHintWin, y = by - kR + i / kHintWin;
            const uint32_t t = *reinterpret_cast<const uint32_t *>(prev + (size_t)clampi(y, 0, H - 1) * (size_t)prevPitch + (size_t)clampi(x, 0, W - 1) * 4u);
            v[k] = (x >= 0 && x < W && y >= 0 && y < H) ? t : 0u;
        }
#pragma unroll
        for (int k = 0; k < kRounds; ++k)
            if (k * kHintThreads + tid < kHintWin * kHintWin) sP[k * kHintThreads + tid] = v[k];
    }
    if (tid < kB * kB) {
        const int x = bx + tid % kB, y = by + tid / kB;
        const bool ok = x >= 0 && x < W && y >= 0 && y < H;
        sC[tid] = ok ? *reinterpret_cast<const uint32_t *>(curr + (size_t)y * (size_t)currPitch + (size_t)x * 4u) : 0u;
        sValid[tid] = ok ? 1u : 0u;
    }
    __syncthreads();
    // Per candidate 64 LDS reads of prev, 64 broadcast reads of curr and 64 v_sad_u8.  A block position outside the
    // image (small frames only: the sample blocks of a frame of 128 x 128 or more lie inside it) has c = 0 and a
    // masked-out prev texel.
    const bool inside = bx >= 0 && by >= 0 && bx + kB <= W && by + kB <= H;          // workgroup-uniform
    uint32_t best = 0xFFFFFFFFu;
    for (int cand = tid; cand < kCand; cand += kHintThreads) {
        const int dyi = cand / kSide, dxi = cand - dyi * kSide;
        const uint32_t *w = sP + dyi * kHintWin + dxi;
        uint32_t sad = 0u;
        if (inside) {
#pragma unroll
            for (int p = 0; p < kB * kB; ++p) sad = __builtin_amdgcn_sad_u8(sC[p], w[(p / kB) * kHintWin + p % kB], sad);
        } else {
#pragma unroll
            for (int p = 0; p < kB * kB; ++p)
                sad = __builtin_amdgcn_sad_u8(sC[p], sValid[p] ? w[(p / kB) * kHintWin + p % kB] : 0u, sad);
        }
        best = min(best, (sad << 11) | (uint32_t)cand);                // <= 64 * 1020 < 2^16, cand < 2^11
    }
    atomicMin(&sBest, best);
    __syncthreads();
    if (tid == 0) hints[blockIdx.x] = sBest;           // best SAD << 11 | candidate (scan index)
}

// (one workgroup; sOwner: lowest hint index that proposes a candidate, sVotes: how many sample blocks propose it -- two LDS
//  atomics per thread.  A version without the two tables, every thread comparing its hint with all 256, was built for the
//  fold into the hint kernel and took 3 us longer.)
__global__ __launch_bounds__(kHints) void motion_order_kernel(
    const uint32_t *__restrict__ hints, const uint32_t *__restrict__ baseScan,
    const uint32_t *__restrict__ entryOfScan, uint32_t *__restrict__ order32) {
    __shared__ uint32_t sOwner[kCand];
    __shared__ uint32_t sVotes[kCand];
    constexpr int kWaves = kHints / 64;
    __shared__ uint32_t sWaveSum[kWaves], sWaveClose[kWaves];
    __shared__ uint32_t sRunning, sTop;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t zero = baseScan[0];               // zero motion
    for (int i = tid; i < kCand; i += kHints) { sOwner[i] = 0xFFFFFFFFu; sVotes[i] = 0u; }
    if (tid == 0) sTop = 0u;
    __syncthreads();
    // Hints are taken in a scrambled order of the sample blocks: under a zoom or a rotation the hints vary smoothly
    // across the frame, and in raster order a pixel would see them approach its own motion -- one running minimum
    // after the other -- which is exactly what fills the lists.
    const uint32_t hint = hints[(tid * 97 + 13) & (kHints - 1)];
    const uint32_t mine = hint & 0x7FFu;
    // A sample block whose best SAD is 1020 or more has no candidate with a cost below 510 (a distance is at least
    // half the sum of its four absolute differences): such a segment would search in full.  Handing segments over
    // (motion_prefilter_kernel) pays when they are the exception; with a quarter of the samples unmatched it is off.
    const uint32_t unmatched = (uint32_t)__popcll(__ballot((hint >> 11) >= 1020u));
    // (close: a match within the lean kernel's reach; the count of EXACT matches rides along in the upper half, for LFG_DEBUG)
    const uint32_t close = (uint32_t)__popcll(__ballot((float)(hint >> 11) < 2.0f * kOnePointOnly)) | ((uint32_t)__popcll(__ballot((hint >> 11) == 0u)) << 16);
    if (lane == 0) { sWaveSum[wv] = unmatched; sWaveClose[wv] = close; }
    if (mine != zero) atomicMin(&sOwner[mine], (uint32_t)tid);
    atomicAdd(&sVotes[mine], 1u);
    __syncthreads();
    // The most popular hint goes first: where it is the answer (a pan: nearly everywhere) the very first evaluation
    // closes the thresholds, and zero motion -- second -- already fails the cheap test instead of being recorded for
    // every pixel.  (Ties: the candidate earlier in scan order.)
    atomicMax(&sTop, (sVotes[mine] << 11) | (uint32_t)(kCand - 1 - (int)mine));
    __syncthreads();
    const uint32_t top = (uint32_t)(kCand - 1) - (sTop & 0x7FFu);
    if (tid == 0) {
        uint32_t unmatchedAll = 0u;
        for (int w = 0; w < kWaves; ++w) unmatchedAll += sWaveSum[w];
        const uint32_t mostMatch = unmatchedAll * 4u <= (uint32_t)kHints ? 1u : 0u;
        order32[kCand] = mostMatch;
        // ... and for the lean kernel (motion_lean.hip), which keeps a segment only while its largest threshold stays below
        // kOnePointOnly: a block whose best SAD is 2 x that or more cannot cost less (a distance is at least half its SAD).  The
        // kernel and the plan that goes with it pay where nearly every sample matches that closely -- a pan +9 %, stills +21 % -- still
        // pay where a few percent of them do not (moving objects, 250 of 256: +5.8 %; occlusions, 242: +0.5 % -- neutral and -1.3 %
        // before the kernel took the rim tiles' inner segments and handed-over segments were searched in four parts) and cost 8 % on
        // frames with sensor noise (none close): the bar is 15 in 16 (63 in 64 until late in round 4).
        uint32_t closeAll = 0u;
        for (int w = 0; w < kWaves; ++w) closeAll += sWaveClose[w];
        // (bit 0: the verdict; above it the two counts, for LFG_DEBUG; bit 31: most sample blocks have a match -- the host sizes the
        //  persistent grid of the lane's next call by it)
        order32[kCand + 2] = (((closeAll & 0xFFFFu) * 16u >= 15u * (uint32_t)kHints) ? 1u : 0u) | ((closeAll & 0xFFFFu) << 1) | (((closeAll >> 16) & 0x7FFFFu) << 12) | (mostMatch << 31);
        order32[0] = entryOfScan[top];
        if (top != zero) order32[1] = entryOfScan[zero];
        sRunning = top != zero ? 2u : 1u;
    }
    __syncthreads();
    // block-wide exclusive scan of one flag per thread, appended at sRunning
    auto append = [&](bool keep, uint32_t scan) {
        const unsigned long long m = __ballot(keep);
        const uint32_t before = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) sWaveSum[wv] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t base = sRunning;
        for (int w = 0; w < wv; ++w) base += sWaveSum[w];
        if (keep) order32[base + before] = entryOfScan[scan];
        __syncthreads();
        if (tid == 0) { uint32_t all = 0u; for (int w = 0; w < kWaves; ++w) all += sWaveSum[w]; sRunning += all; }
        __syncthreads();
    };
    constexpr int kRounds = (kCand - 1 + kHints - 1) / kHints;
    uint32_t base[kRounds];                                            // this thread's entries of the fixed order, read up front
#pragma unroll
    for (int k = 0; k < kRounds; ++k) base[k] = baseScan[min(1 + k * kHints + tid, kCand - 1)];
    append(mine != zero && mine != top && sOwner[mine] == (uint32_t)tid, mine);       // the other distinct hints
    if (tid == 0) order32[kCand + 1] = sRunning;                       // entries in front: top hint, zero motion, the other hints
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {                                // then everything no hint proposed
        const int e = 1 + k * kHints + tid;
        append(e < kCand && sOwner[base[k]] == 0xFFFFFFFFu, base[k]);
    }
}

hipError_t launch_motion_prefiltered_8_16(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                          const lfg_frame &mv, uint8_t *workspace, const MotionWorkspaceLayout &l, int units,
                                          const uint32_t *rank2scan, const uint32_t *order,
                                          const uint32_t *entryOfScan, const uint32_t *baseScan, bool useHints, bool framesInFlight,
                                          const FusedOut &fused, bool lean, uint32_t *leanFlagHost, int groupsCap, bool expectNoFallback) {
    const int tilesX = ((int)curr.width + kTW - 1) / kTW;
    lean = lean && useHints && !fused.data && l.units2 > 0 && l.leanCount > 0 && curr.width >= 64u && curr.height >= 64u && lean_frames_ok(prev, curr, mv);
    if (lean) units = l.units2;          // the plan that goes with the lean kernel
    Rec *list = reinterpret_cast<Rec *>(workspace + l.list);
    float *umin = reinterpret_cast<float *>(workspace + l.umin);
    uint32_t *count = reinterpret_cast<uint32_t *>(workspace + l.count);
    uint32_t *flags = reinterpret_cast<uint32_t *>(workspace + l.tileFlags);
    static_assert(kPTH == kTH, "prefilter tiles and exact tiles share their rows");
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
    sp.openCount = reinterpret_cast<uint32_t *>(workspace + l.ctrl) + 4;
    sp.fused = fused;
    sp.dynParts = framesInFlight ? 4 : LFG_DYN_PARTS;
    if (const char *dp = getenv("LFG_DYN_PARTS_RT")) sp.dynParts = atoi(dp) == 4 ? 4 : LFG_DYN_PARTS;      // (measurement)
    // (as many SEGMENTS as the queue holds in LFG_DYN_PARTS parts, whatever the parts of this call: one block of thresholds each)
    sp.queueCap = l.queueCap / ((LFG_DYN_PARTS / 4) / (sp.dynParts / 4));
    sp.unitsStatic = lean ? l.units2Static : units;
    sp.hardTiles = lean ? reinterpret_cast<const uint32_t *>(workspace + l.hardTiles) : nullptr;
    sp.hardCount = lean ? reinterpret_cast<const uint32_t *>(workspace + l.ctrl) + 5 : nullptr;          // (ctrl[5]: cleared by the hint kernel with the rest)
    uint32_t *segDone = reinterpret_cast<uint32_t *>(workspace + l.segDone);
    uint32_t *const ctrl = reinterpret_cast<uint32_t *>(workspace + l.ctrl);
    hipError_t e = hipSuccess;
    uint32_t *verdictWord = nullptr;       // this call's verdict word (its own order table's entry kCand + 2), where there is one
    if (useHints && curr.width >= 64u && curr.height >= 64u) {
        // this call's visiting order (motion_hint_kernel, which also clears the call's control area -- tile flags, segment
        // marks and map, counters, queue -- and motion_order_kernel)
        uint32_t *hints = reinterpret_cast<uint32_t *>(workspace + l.order);
        uint32_t *callOrder = hints + kHints;
        // (256 threads a workgroup beside the prefilter launches of other frames in flight -- see the kernel --, 1,024 alone)
        if (framesInFlight)
            hipLaunchKernelGGL(motion_hint_kernel<kHintThreadsInFlight>, dim3(kHints), dim3(kHintThreadsInFlight), 0, s,
                               (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                               (int)curr.width, (int)curr.height, hints, flags, (int)((l.order - l.tileFlags) / sizeof(uint32_t)));
        else
            hipLaunchKernelGGL(motion_hint_kernel<kHintThreadsAlone>, dim3(kHints), dim3(kHintThreadsAlone), 0, s,
                               (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                               (int)curr.width, (int)curr.height, hints, flags, (int)((l.order - l.tileFlags) / sizeof(uint32_t)));
        hipLaunchKernelGGL(motion_order_kernel, dim3(1), dim3(kHints), 0, s, hints, baseScan, entryOfScan, callOrder);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        order = callOrder;
        verdictWord = callOrder + kCand + 2;
        // the whole interior tiles, through the lean kernel first (motion_lean.hip): what it settles it marks in segDone -- cleared by
        // the hint kernel above -- and the generic kernel below skips
        if (lean) {
            // (beside the persistent kernel on a stream of its own, with that kernel's grid cut to 448 .. 320 workgroups to leave it room:
            //  measured for a context that runs one frame at a time, 2,160 - 2,430 frames/s against 2,610 without the kernel: not done)
            e = launch_motion_lean(s, prev, curr, mv, order, reinterpret_cast<const uint32_t *>(workspace + l.leanTiles), l.leanLaunch, sp.tilesX, segDone,
                                   reinterpret_cast<uint32_t *>(workspace + l.hardTiles), ctrl + 5, ctrl + 6,
                                   getenv("LFG_LEAN_FORCE") != nullptr && atoi(getenv("LFG_LEAN_FORCE")) == 1);      // (ctrl[5]: tiles it left; ctrl[6], [7]: segments it settled, segments it left -- counted in diagnostic builds)
            if (e != hipSuccess) return e;
        }
    } else {
        e = hipMemsetAsync(flags, 0, l.order - l.tileFlags, s);         // (no hints: the fixed order; the area is cleared by a memset)
        if (e != hipSuccess) return e;
    }
    int groups = std::max(1, std::min(sp.units, l.slots > 0 ? l.slots : sp.units));
    if (groupsCap > 0) groups = std::max(1, std::min(groups, groupsCap));      // (frames in flight: lfg_capi.cpp, motion_run)
    if (const char *g = getenv("LFG_PREF_GROUPS")) groups = std::max(1, std::min(l.slots > 0 ? l.slots : groups, atoi(g)));      // (measurement)
    if (fused.data)
        hipLaunchKernelGGL(motion_prefilter_kernel<true>, dim3(groups), dim3(kPNT), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int)curr.width, (int)curr.height, list, umin, count, flags, tilesX, order, sp,
                           (int8_t *)mv.data, (int)mv.pitch, rank2scan, segDone, ctrl);
    else
        hipLaunchKernelGGL(motion_prefilter_kernel<false>, dim3(groups), dim3(kPNT), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int)curr.width, (int)curr.height, list, umin, count, flags, tilesX, order, sp,
                           (int8_t *)mv.data, (int)mv.pitch, rank2scan, segDone, ctrl);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
#ifdef LFG_DIAG_NO_SLOW
    return hipSuccess;                 // (timing experiment: the prefilter alone; its output is not fit for the resolve kernel)
#endif
#ifdef LFG_MOTION_STAMPS
    {
        static int calls = 0;
        if (++calls == 3) {
            std::vector<unsigned long long> h(8192 * 32);
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(gMotionStamps), h.size() * 8);
            const int n = std::min(sp.units, 8192);
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int u = 0; u < n; ++u) for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1]) continue; t0 = std::min(t0, o[0]); t1 = std::max(t1, o[1]); }
            double sum[2] = {0, 0}, mx[2] = {0, 0}, ev[2] = {0, 0}, stg[2] = {0, 0}, fst[2] = {0, 0}, bat[2] = {0, 0}, srch[2] = {0, 0}; int cnt[2] = {0, 0};
            for (int u = 0; u < n; ++u) {
                unsigned long long a = ~0ull, b = 0, evals = 0, batches = 0; int rim = 0; double st = 0, fs = 0, se = 0;
                for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1]) continue; a = std::min(a, o[0]); b = std::max(b, o[1]); evals += o[2]; batches += o[3] & 0xFFFFFFFFull; rim = (int)((o[3] >> 32) & 1); se = std::max(se, (double)(o[3] >> 33) / 100.0);
                    st = std::max(st, (double)(o[4] - o[0]) / 100.0); if (o[5]) fs = std::max(fs, (double)(o[5] - o[4]) / 100.0); }
                if (!b) continue;
                const double us = (double)(b - a) / 100.0;
                sum[rim] += us; mx[rim] = std::max(mx[rim], us); ev[rim] += (double)evals / 4; bat[rim] += (double)batches / 4; srch[rim] += se; stg[rim] += st; fst[rim] += fs; ++cnt[rim];
                if (u % 197 == 0) fprintf(stderr, "unit %d rim %d start %.1f us dur %.1f us evals/wave %.1f\n", u, rim, (double)(a - t0) / 100.0, us, (double)evals / 4);
            }
#if defined(LFG_STAMP_LATTICE3) || defined(LFG_STAMP_LATTICE4)
            {
                double t[4] = {0, 0, 0, 0}; int nwv = 0;
                for (int u = 0; u < n; ++u) for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1] || ((o[3] >> 32) & 1) || !o[7]) continue; for (int k = 0; k < 4; ++k) t[k] += (double)((o[7] >> (16 * k)) & 0xFFFF) / 100.0; ++nwv; }
                fprintf(stderr, "interior waves %d, after the staging barrier: batch loop entered %.2f us, first candidate chosen %.2f, evaluated %.2f, thresholds reduced %.2f\n", nwv, t[0] / std::max(nwv, 1), t[1] / std::max(nwv, 1), t[2] / std::max(nwv, 1), t[3] / std::max(nwv, 1));
            }
#elif defined(LFG_STAMP_LATTICE)
            {
                double sk = 0, fl = 0; int nwv = 0;
                for (int u = 0; u < n; ++u) for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1] || ((o[3] >> 32) & 1)) continue; const unsigned box = (unsigned)o[7]; sk += box & 0xFFFF; fl += box >> 16; ++nwv; }
                fprintf(stderr, "interior waves %d: lookahead passes that skipped %.2f, that did not %.2f per wave\n", nwv, sk / std::max(nwv, 1), fl / std::max(nwv, 1));
            }
#endif
            for (int r = 0; r < 2; ++r)
                fprintf(stderr, "%s units %d: mean %.1f us (staging %.1f, first batch %.1f, search over after %.1f), max %.1f us, evaluations per wave %.1f, batches per wave %.1f\n", r ? "rim" : "interior", cnt[r],
                        sum[r] / std::max(cnt[r], 1), stg[r] / std::max(cnt[r], 1), fst[r] / std::max(cnt[r], 1), srch[r] / std::max(cnt[r], 1), mx[r], ev[r] / std::max(cnt[r], 1), bat[r] / std::max(cnt[r], 1));
            fprintf(stderr, "span %.1f us\n", (double)(t1 - t0) / 100.0);
            {   // handed-over segments (queue units follow the plan's units in the stamp array)
                unsigned long long qa = ~0ull, qb = 0, qlast = 0; double qsum = 0; int qn = 0; double qev = 0;
                for (int u = n; u < 8192; ++u) for (int w = 0; w < 4; ++w) {
                    const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1] || !((o[6] >> 9) & 1)) continue;
                    qa = std::min(qa, o[0]); qlast = std::max(qlast, o[0]); qb = std::max(qb, o[1]); qsum += (double)(o[1] - o[0]) / 100.0; qev += (double)o[2]; ++qn;
                    t1 = std::max(t1, o[1]);
                }
                {   // handed-over segments (two consecutive queue units = eight parts) in which some part ended with every pixel at
                    // a zero-cost candidate: one evaluation of the right candidate would have settled the whole segment
                    int segs = 0, zero = 0, narrowSegs = 0, bandSegs = 0;
                    for (int u = n; u + 1 < 8192; u += 2) {
                        bool any = false, z = false, nar = false, bnd = false;
                        for (int k = 0; k < 8; ++k) { const unsigned long long *o = &h[((u + k / 4) * 4 + k % 4) * 8]; if (!o[1]) continue; any = true;
                            const uint32_t tb = (uint32_t)(o[7] >> 32); float fb; memcpy(&fb, &tb, 4); if (fb < 0.5f) z = true;
                            const int kk = (int)(o[6] & 7u); if (kk == 7) bnd = true; else if (kk) nar = true; }
                        if (any) { ++segs; zero += z; narrowSegs += nar; bandSegs += bnd; }
                    }
                    fprintf(stderr, "handed-over segments %d: %d end with a zero-cost candidate for every pixel (a hint would have settled them), %d searched narrow, %d as a row band\n", segs, zero, narrowSegs, bandSegs);
                }
                if (qn) fprintf(stderr, "queue units: %d waves, first start %.1f, last start %.1f, last end %.1f us, mean duration %.1f us, evaluations per wave %.1f\n", qn,
                                (double)(qa - t0) / 100.0, (double)(qlast - t0) / 100.0, (double)(qb - t0) / 100.0, qsum / qn, qev / qn);
                unsigned long long pe = 0; for (int u = 0; u < n; ++u) for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (o[1]) pe = std::max(pe, o[1]); }
                fprintf(stderr, "plan units: last end %.1f us\n", (double)(pe - t0) / 100.0);
            }
            {   // the waves that finish last
                std::vector<std::pair<unsigned long long, int>> ends;
                for (int u = 0; u < n; ++u) for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (o[1]) ends.push_back({o[1], u * 4 + w}); }
                std::sort(ends.begin(), ends.end());
                for (size_t i = ends.size() > 14 ? ends.size() - 14 : 0; i < ends.size(); ++i) {
                    const unsigned long long *o = &h[(size_t)ends[i].second * 8];
                    fprintf(stderr, "  late: unit %d wave %d tile (%d,%d) seg %d start %.1f end %.1f us, evals %llu, batches %llu, staged after %.1f us\n", ends[i].second / 4, ends[i].second % 4,
                            (int)((o[6] >> 12) & 0xFF), (int)((o[6] >> 20) & 0xFF), (int)((o[6] >> 10) & 3), (double)(o[0] - t0) / 100.0, (double)(o[1] - t0) / 100.0, o[2], o[3] & 0xFFFFFFFFull, (double)(o[4] - o[0]) / 100.0);
                }
            }
            {   // rim units by where they lie: mean duration, evaluations and batches per wave
                const char *names[6] = {"top", "bottom, last segment", "bottom, other segments", "left", "right", "corner or other"};
                double dur[6] = {0}, evs[6] = {0}, bts[6] = {0}, thrA[6] = {0}, thrB[6] = {0}; int nu[6] = {0}, nNarrow[6] = {0}, nWaves[6] = {0}, nZero[6] = {0}, nBand[6] = {0}; double wBand[6] = {0}, srchK[6] = {0}, fstK[6] = {0};
                double hardDur[6] = {0}, hardEv[6] = {0}, hardBt[6] = {0}, hardPh[6][3] = {{0}};
                double inDur = 0, inEv = 0, inPh[3] = {0, 0, 0}; int inN = 0;
                int maxTx = 0, maxTy = 0;
                for (int u = 0; u < n; ++u) { const unsigned long long *o = &h[(u * 4) * 8]; if (!o[1]) continue; maxTx = std::max(maxTx, (int)((o[6] >> 12) & 0xFF)); maxTy = std::max(maxTy, (int)((o[6] >> 20) & 0xFF)); }
                for (int u = 0; u < n; ++u) {
                    unsigned long long a = ~0ull, b = 0, evals = 0, batches = 0; int rim = 0, tx = 0, ty = 0, sg = 0, nw = 0; double se = 0, fs = 0;
                    for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1]) continue; a = std::min(a, o[0]); b = std::max(b, o[1]); evals += o[2]; batches += o[3] & 0xFFFFFFFFull;
                        se = std::max(se, (double)(o[3] >> 33) / 100.0); if (o[5]) fs = std::max(fs, (double)(o[5] - o[0]) / 100.0);
                        rim = (int)((o[3] >> 32) & 1); tx = (int)((o[6] >> 12) & 0xFF); ty = (int)((o[6] >> 20) & 0xFF); sg = (int)((o[6] >> 10) & 3); ++nw; }
#ifdef LFG_STAMP_PHASES
                    if (b && !rim) for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1]) continue; ++inN; inDur += (double)(o[1] - o[0]) / 100.0; inEv += (double)(o[2] & 0xFFFFull);
                        inPh[0] += (double)(o[5] & 0xFFFFFull) / 100.0; inPh[1] += (double)((o[5] >> 20) & 0xFFFFFull) / 100.0; inPh[2] += (double)((o[5] >> 40) & 0xFFFFFull) / 100.0; }
#endif
                    if (!b || !rim) continue;
                    const bool l = tx == 0, r = tx == maxTx, t = ty == 0, bo = ty == maxTy;
                    const int k = (l + r + t + bo) != 1 ? 5 : t ? 0 : bo ? (sg == 2 ? 1 : 2) : l ? 3 : 4;
                    dur[k] += (double)(b - a) / 100.0; evs[k] += (double)evals / nw; bts[k] += (double)batches / nw; ++nu[k]; srchK[k] += se; fstK[k] += fs;
                    for (int w = 0; w < 4; ++w) { const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1]) continue; ++nWaves[k]; if (o[6] & 7u) ++nNarrow[k];
                        const uint32_t ta = (uint32_t)o[7], tb = (uint32_t)(o[7] >> 32); float fa, fb; memcpy(&fa, &ta, 4); memcpy(&fb, &tb, 4);
                        if (fa < 0.5f) ++nZero[k]; else { thrA[k] += std::min(fa, 1e5f); thrB[k] += std::min(fb, 1e5f);
                            hardDur[k] += (double)(o[1] - o[0]) / 100.0; hardEv[k] += (double)(o[2] & 0xFFFFull); hardBt[k] += (double)(o[3] & 0xFFFFFFFFull);
#ifdef LFG_STAMP_PHASES
                            hardPh[k][0] += (double)(o[5] & 0xFFFFFull) / 100.0; hardPh[k][1] += (double)((o[5] >> 20) & 0xFFFFFull) / 100.0; hardPh[k][2] += (double)((o[5] >> 40) & 0xFFFFFull) / 100.0;
#endif
                        }
                        const unsigned box = (unsigned)(o[6] >> 32); if (((box >> 16) & 0xFF) == 0xEE) { ++nBand[k]; wBand[k] += (double)(((box >> 8) & 0xFF) - (box & 0xFF) + 1); } }
                }
                for (int k = 0; k < 6; ++k) if (nu[k]) fprintf(stderr, "rim units, %s: %d, mean %.1f us (first batch done after %.1f, search over after %.1f), evaluations per wave %.1f, batches per wave %.1f; waves %d, narrow %d, all-zero thresholds after the hints %d, others' mean threshold after hints %.0f, at the end %.0f; band known in %d waves, mean width %.1f columns\n", names[k], nu[k], dur[k] / nu[k], fstK[k] / nu[k], srchK[k] / nu[k], evs[k] / nu[k], bts[k] / nu[k],
                                                        nWaves[k], nNarrow[k], nZero[k], thrA[k] / std::max(1, nWaves[k] - nZero[k]), thrB[k] / std::max(1, nWaves[k] - nZero[k]), nBand[k], wBand[k] / std::max(1, nBand[k]));
                for (int k = 0; k < 6; ++k) if (nWaves[k] - nZero[k] > 0) fprintf(stderr, "  %s, the %d waves that are not settled by the hints: mean %.1f us, %.1f full evaluations, %.1f batches\n", names[k], nWaves[k] - nZero[k],
                                                        hardDur[k] / (nWaves[k] - nZero[k]), hardEv[k] / (nWaves[k] - nZero[k]), hardBt[k] / (nWaves[k] - nZero[k]));
#ifdef LFG_STAMP_PHASES
                if (inN) fprintf(stderr, "  interior units, their %d waves: mean %.1f us, %.1f full evaluations; %.1f us in the batch tests, %.1f in the eight- and sixteen-point tests, %.1f in full evaluations\n",
                                 inN, inDur / inN, inEv / inN, inPh[0] / inN, inPh[1] / inN, inPh[2] / inN);
                for (int k = 0; k < 6; ++k) if (nWaves[k] - nZero[k] > 0) fprintf(stderr, "  %s, those waves: %.1f us in the batch tests, %.1f in the sixteen-point test, %.1f in full evaluations\n", names[k],
                                                        hardPh[k][0] / (nWaves[k] - nZero[k]), hardPh[k][1] / (nWaves[k] - nZero[k]), hardPh[k][2] / (nWaves[k] - nZero[k]));
#endif
            }
            // (inside the block above: the waves whose thresholds are not all zero-cost words after the hints)
            {   // narrow search: waves by candidates per pass, their mean duration
                double dur[8] = {0}; int nw[8] = {0}; double evs[8] = {0};
                for (int u = 0; u < 8192; ++u) for (int w = 0; w < 4; ++w) {
                    const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1]) continue;
                    const int k = (int)(o[6] & 7u); if (!k) continue;
                    dur[k] += (double)(o[1] - o[0]) / 100.0; evs[k] += (double)o[2]; ++nw[k];
                }
                for (int k = 1; k < 8; ++k) if (nw[k]) fprintf(stderr, "narrow search, %d candidates per pass: %d waves, mean %.1f us, wide evaluations per wave %.1f\n", k, nw[k], dur[k] / nw[k], evs[k] / nw[k]);
            }
            {   // full-search waves (plan and queue units) by the box of their pixels without a match
                int byRuns[9] = {0}, half[3] = {0}, total = 0;
                for (int u = 0; u < 8192; ++u) for (int w = 0; w < 4; ++w) {
                    const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1] || o[2] <= 100) continue;
                    const unsigned box = (unsigned)(o[6] >> 32); const int lo = box & 0xFF, hi = (box >> 8) & 0xFF, rlo = (box >> 16) & 0xFF, rhi = (box >> 24) & 0xFF;
                    ++total;
                    if (hi < lo || lo > 8) { ++byRuns[0]; continue; }
                    ++byRuns[std::min(8, hi - lo + 1)];
                    if (rhi < 8) ++half[0]; else if (rlo >= 8) ++half[1]; else ++half[2];
                }
                fprintf(stderr, "full-search waves %d; runs of 7 columns that hold unmatched pixels: none/unknown %d, 1: %d, 2: %d, 3: %d, 4: %d, 5: %d, 6: %d, 7: %d, 8: %d; rows 0-7 only %d, rows 8-15 only %d, both %d\n",
                        total, byRuns[0], byRuns[1], byRuns[2], byRuns[3], byRuns[4], byRuns[5], byRuns[6], byRuns[7], byRuns[8], half[0], half[1], half[2]);
            }
            {   // a sample of the waves that evaluated more than 100 candidates in full: where, and against which thresholds
                int shown = 0;
                for (int u = 0; u < n; ++u) for (int w = 0; w < 4; ++w) {
                    const unsigned long long *o = &h[(u * 4 + w) * 8]; if (!o[1]) continue;
                    if (o[2] > 100 && (u % 7) == 0 && w == 1 && shown < 24) { ++shown;
                        const uint32_t ta = (uint32_t)o[7], tb = (uint32_t)(o[7] >> 32); float fa, fb; memcpy(&fa, &ta, 4); memcpy(&fb, &tb, 4);
                        fprintf(stderr, "  tile (%d,%d) seg %d: unit %d wave %d evals %llu dur %.1f us  waveThr after hints %.1f, at last batch %.1f, four-point still on %d\n", (int)((o[6] >> 12) & 0xFF), (int)((o[6] >> 20) & 0xFF), (int)((o[6] >> 10) & 3), u, w, o[2], (double)(o[1] - o[0]) / 100.0, fa, fb, (int)((o[6] >> 8) & 1)); }
                }
            }
            unsigned long long rs[12];
            hipMemcpyFromSymbol(rs, HIP_SYMBOL(gResolveStats), sizeof(rs));
            fprintf(stderr, "resolve (previous calls together): %llu working waves, mean gather %.2f us, mean rest %.2f us, longest wave %.1f us, "
                            "%llu pixels taken together, first start to last end (all calls and what lies between) %.1f us\n",
                    rs[4], rs[4] ? rs[5] / 100.0 / rs[4] : 0.0, rs[4] ? rs[6] / 100.0 / rs[4] : 0.0, rs[7] / 100.0, rs[8], (rs[10] - rs[9]) / 100.0);
            fprintf(stderr, "resolve (previous calls together): %llu pixels with several survivors (%llu within 64 px of the rim), %llu survivors\n", rs[0], rs[2], rs[1]);
        }
    }
#endif
    const int segments = sp.tilesX * (((int)curr.height + kPTH - 1) / kPTH) * (kPTH / kSeg);
    // (a call that went through the lean kernel -- a pan, stills, moving objects -- takes the small grid with frames in flight too:
    //  pan +2.1 %, stills +3 %, occluded and moving objects +0.8 %; noise, which does not, loses 1.8 % by it and keeps the large one)
    int resolveGroups = (framesInFlight && !lean) ? segments * (kSeg / 4) : std::min(kResolveGroups, segments * (kSeg / 4));
    if (const char *rg = getenv("LFG_RESOLVE_GROUPS")) resolveGroups = std::max(1, std::min(segments * (kSeg / 4), atoi(rg)));      // (measurement)
    if (fused.data)
        hipLaunchKernelGGL(motion_resolve_kernel<true>, dim3((unsigned)std::max(1, resolveGroups)), dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height, list, umin, count, flags, tilesX, sp, rank2scan, segDone);
    else
        hipLaunchKernelGGL(motion_resolve_kernel<false>, dim3((unsigned)std::max(1, resolveGroups)), dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height, list, umin, count, flags, tilesX, sp, rank2scan, segDone);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
#ifdef LFG_DIAG_NO_FALLBACK             // (timing experiment: what the launch that usually finds nothing to do costs)
    return hipSuccess;
#endif
    e = launch_motion_tiled_8_16(s, prev, curr, mv, flags, rank2scan, reinterpret_cast<unsigned long long *>(workspace + l.merge),
                                 sp.queueCount + 1, fused, expectNoFallback && verdictWord != nullptr, verdictWord,
                                 verdictWord ? leanFlagHost : nullptr);
    // (the order kernel's verdict on this call's content (order32[kCand + 2]) reaches the host -- which decides with it how the lane's
    //  NEXT call is launched -- by a store of that last launch into the host's pinned word; until round 4's end a copy command
    //  behind the call, a dispatch of its own on the lane's stream: without it the pan is where it was, stills, moving objects and
    //  occlusions +0.5 %)
    return e;
}

// Diagnostic: compares exact_sqrt with __builtin_sqrtf for every float whose bit pattern lies in
// [lo_bits, hi_bits]; counts mismatches.  Used only by the test-suite (lfg_selftest_sqrt).
__global__ __launch_bounds__(256) void sqrt_selftest_kernel(uint32_t lo, uint32_t hi, unsigned long long *mismatch) {
    const unsigned long long span = (unsigned long long)hi - lo + 1ull;
    unsigned long long bad = 0;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < span; i += (unsigned long long)gridDim.x * 256ull) {
        const float x = __builtin_bit_cast(float, (uint32_t)(lo + i));
        const float a = exact_sqrt(x), b = __builtin_sqrtf(x);
        if (__builtin_bit_cast(uint32_t, a) != __builtin_bit_cast(uint32_t, b)) ++bad;
    }
    if (bad) atomicAdd(mismatch, bad);
}

hipError_t launch_sqrt_selftest(hipStream_t s, uint32_t lo, uint32_t hi, unsigned long long *d_mismatch) {
    hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(4096), dim3(256), 0, s, lo, hi, d_mismatch);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ generic (literal)

template <bool INTENDED>     // tie-break: false = motion.comp (first in scan order), true = shortest vector, then scan order
__global__ __launch_bounds__(256) void motion_generic_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H, int B, int R) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    const int bsx = px - B / 2, bsy = py - B / 2;
    float minDiff = 1e10f;
    int bx = 0, by = 0, bestD2 = 0x7FFFFFFF;
    for (int dy = -R; dy <= R; ++dy) {
        for (int dx = -R; dx <= R; ++dx) {
            float diff = 0.0f;
            for (int y = 0; y < B; ++y) {
                const int cy = bsy + y;
                if (cy < 0 || cy >= H) continue;
                for (int x = 0; x < B; ++x) {
                    const int cx = bsx + x;
                    if (cx < 0 || cx >= W) continue;
                    const uint32_t c = *reinterpret_cast<const uint32_t *>(curr + (size_t)cy * (size_t)currPitch + (size_t)cx * 4u);
                    const int qx = cx + dx, qy = cy + dy;
                    uint32_t p = 0u;
                    if (qx >= 0 && qy >= 0 && qx < W && qy < H)
                        p = *reinterpret_cast<const uint32_t *>(prev + (size_t)qy * (size_t)prevPitch + (size_t)qx * 4u);
                    const float cc[4] = {unorm8_to_float(byte0(c)), unorm8_to_float(byte1(c)),
                                         unorm8_to_float(byte2(c)), unorm8_to_float(byte3(c))};
                    const f32x4 pp = {unorm8_to_float(byte0(p)), unorm8_to_float(byte1(p)),
                                      unorm8_to_float(byte2(p)), unorm8_to_float(byte3(p))};
                    diff += dist4<false>(cc, pp);
                }
            }
            if (INTENDED) {
                const int d2 = dx * dx + dy * dy;
                if (diff < minDiff || (diff == minDiff && d2 < bestD2)) { minDiff = diff; bestD2 = d2; bx = dx; by = dy; }
            } else if (diff < minDiff) { minDiff = diff; bx = dx; by = dy; }
        }
    }
    int8_t *dst = mv + (size_t)py * (size_t)mvPitch + (size_t)px * 2u;
    dst[0] = (int8_t)bx; dst[1] = (int8_t)by;
}

hipError_t launch_motion_generic(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                 const lfg_frame &mv, int block_size, int radius, bool intended) {
    dim3 grid((curr.width + 63) / 64, (curr.height + 3) / 4);
    if (intended)
        hipLaunchKernelGGL(motion_generic_kernel<true>, grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height, block_size, radius);
    else
        hipLaunchKernelGGL(motion_generic_kernel<false>, grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height, block_size, radius);
    return hipGetLastError();
}

}  // namespace lfg
