// motion_literal.hip -- per-pixel full-search block matching, the MI355X-native replacement of
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
// (The prefiltered path that produces the same vectors without running this chain for every candidate: motion_prefilter.hip and
//  the files listed in lfg_motion_common.hpp.  This file is what they are checked against, and what flagged tiles go through.)
#include "lfg_motion_common.hpp"

#include <algorithm>

namespace lfg {

// ------------------------------------------------------------------------------ tiled, B = 8, R = 16

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

// The second pass's grid when the lane's previous call flagged nothing (motion_tiled_8_16_loop_kernel).  What it buys in the
// steady state is small and falls with its size (64 workgroups: +1.8 % frames/s under the pan, 256: +0.6 %, round 4); what it
// costs is the call in which the prediction is WRONG -- the first frame of a fade onto flat content, a cut: a part of a flagged
// tile is 0.54 ms of one workgroup, a whole tile 4.3 ms, so 100 flagged tiles are 6.7 ms on 64 workgroups, 1.7 ms on 256 and
// 1.1 ms on the full grid; a fully flagged 4K frame 137 / 34 / 20 ms.  256 bounds the hitch at 1.7 x the full grid's (round 4
// shipped 64: a multi-frame stall on a real-time path for 1.2 % more).  A multiple of eight (fallback_item).
constexpr int kFallbackSmallGrid = 256;

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

// tileFlags == nullptr: the literal kernel for every tile.  Otherwise the second pass of the prefiltered path: one launch,
// whatever was flagged (see the kernel).
hipError_t launch_motion_tiled_8_16(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                    const lfg_frame &mv, const uint32_t *tileFlags, const uint32_t *rank2scan,
                                    unsigned long long *merge, uint32_t *flaggedTiles, const FusedOut &fused,
                                    bool expectNothing, uint32_t *verdictWord, uint32_t *hostWord) {
    const int tilesX = ((int)curr.width + kTW - 1) / kTW, tilesY = ((int)curr.height + kTH - 1) / kTH;
    // (expectNothing: the lane's previous call flagged no tile -- kFallbackSmallGrid looping workgroups, which take it all if this
    //  one does after all)
    const dim3 grid = tileFlags ? dim3(expectNothing ? (unsigned)kFallbackSmallGrid : (unsigned)std::max(tilesX * tilesY, kShareBelow * kFallbackParts), 1, 1)
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
