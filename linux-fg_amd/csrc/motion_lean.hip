// motion_lean.hip -- the prefilter of shaders/motion.comp:27-52 for the tiles that need almost none of it.
//
// Under a pan, or where nothing moves, nine work units in ten of the prefiltered motion path (csrc/motion_prefilter.hip) are whole tiles
// away from the image's rim whose every pixel finds its answer in the call's first hint: one evaluation (or none: the hint reads
// the very bytes of the current frame), after which every other candidate fails the cheapest test there is -- one texel compare
// at each of the 14 lattice points that put a point into every pixel's 8 x 8 block (lfg_motion_tile.hpp).  The generic persistent
// kernel serves those tiles with the same 1,800-line unit that serves rim segments, parts of tiles and the hand-over queue: 256
// VGPRs, 75 KB of LDS, two workgroups per CU, and a chain of latencies per unit that two waves per SIMD cannot hide.
//
// This kernel is that path WRITTEN as the small kernel it is: no lists, no records, no narrow search, no plateaus (the search
// window of its tiles lies inside prev), no queue, no parts of the order.  A wave owns a 16-row segment of a 56 x 64 tile as in
// the generic kernel and keeps, per pixel, what the bracket argument needs and nothing else:
//     thr   kRatio x the smallest S~ seen (or, for a candidate whose S~ is 0 and whose cost therefore is exactly 0, the smallest
//           such rank, carried inside the word: prefilter_records.inc, "zeroCap"),
//     best  the rank of the candidate that set it,
//     amb   "a second candidate came within the bracket of the best one" -- only the literal 64-term chain can decide between two
//           such candidates, and this kernel has no business there.
// S~ is the generic kernel's bracket value -- exact integer squared distances, v_sqrt_f32, a depth-6 pairwise tree: within 9 u of
// 255 x the shader's real-arithmetic cost -- so "the only evaluated candidate with S~ <= kRatio min S~" IS the shader's first
// strict minimum (motion_prefilter.hip, "Bracket"); every candidate that is not evaluated in full has a lattice distance above the wave's
// largest threshold and S~, a rounded sum of non-negative distances, is never below any of its terms.
// A segment that does not fit the pattern -- its largest threshold after the hints is not small (noise, a moving object's rim,
// an occlusion: the four- and sixteen-point tests, the narrow search and the hand-over are the generic kernel's), or two
// candidates tie within the bracket -- is simply LEFT: the wave stops, nothing of it is written, and the generic kernel, which
// skips the segments marked done here, does it from scratch.  Results therefore never depend on this kernel; only the time does.
//
// 154 VGPRs and 48 KB of LDS (50 KB under the intended tie order: its rank table): three workgroups per CU.
#include "lfg_device.hpp"
#include "lfg_internal.hpp"
#include "lfg_motion_tile.hpp"

namespace lfg {

typedef const __attribute__((address_space(3))) uint32_t *lean_lds_u32;
typedef const __attribute__((address_space(3))) float *lean_lds_f32;

constexpr int kLeanSlabP = 68;                     // slab row pitch (floats): = 4 mod 32, so the transposed reads of 8 rows x 4 runs of a half wave
                                                  // (address r8 * 68 + 7 q + i) fall into 32 distinct banks; the writes are consecutive
constexpr int kLeanSlab = 8 * kLeanSlabP;          // eight rows at a time: 2,176 bytes per wave
static_assert((kCand + 31) / 32 <= 64 && kLeanSlabP % 32 == 4 && kLeanSlabP >= kPTW + kB - 1 && kLeanSlab * 4 >= kSeg * kPTW * 2, "conflict-free, and room for a segment's vectors");
#ifndef LFG_LEAN_AHEAD
#define LFG_LEAN_AHEAD 3                          // candidates per lane and pass over the ranks
#endif
constexpr int kLeanAhead = LFG_LEAN_AHEAD;
constexpr int kLeanHintsMax = 64;                  // entries of the call's order taken as hints (the generic kernel's limit): one per lane
#ifndef LFG_LEAN_EVALS_MAX
#define LFG_LEAN_EVALS_MAX 24                      // full evaluations after which a segment is not the easy kind: left to the generic kernel
#endif

struct LeanState {
    f32x2 thr[kRun];           // (row r8, row r8 + 8) x 7 pixels: see the head of this file
    uint32_t best[kRun];       // rank of the pixel's best candidate: row r8 in the low half, row r8 + 8 in the high half
    uint32_t amb;              // bit i: pixel i of row r8, bit 16 + i: of row r8 + 8
};

// n = |c|^2 + |p|^2 - 2 c.p as a float without conversions (prefilter_sums.inc: columnSums): the dot products accumulate onto float bit
// patterns, 0x4B000000 + k = 2^23 + k and 0x4B800000 + k = 2^24 + 2 k.
__device__ __forceinline__ float lean_sqdist(uint32_t c, uint32_t cc, uint32_t p) {
    const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(p, p, cc, false));
    const float f2 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(c, p, 0x4B800000u, false));
    return (f1 - f2) + 8388608.0f;
}

// The 23 distances of a position column and their sixteen sums over eight rows, C8[j] = (V8 of row j, V8 of row j + 8): the
// generic kernel's shared pairwise tree on packed pairs (position j with position j + 8).
__device__ __forceinline__ void lean_column_sums(const uint32_t (&p)[kSegD], const uint32_t (&c)[kSegD], const uint32_t (&cc)[kSegD], f32x2 (&C8)[8]) {
    constexpr int kPairs = kSegD - 8;                                  // 15
    f32x2 A[kPairs];
#pragma unroll
    for (int j = 0; j < 8; ++j) A[j] = f32x2{__builtin_amdgcn_sqrtf(lean_sqdist(c[j], cc[j], p[j])), __builtin_amdgcn_sqrtf(lean_sqdist(c[j + 8], cc[j + 8], p[j + 8]))};
#pragma unroll
    for (int j = 8; j < kPairs; ++j) A[j] = f32x2{A[j - 8].y, __builtin_amdgcn_sqrtf(lean_sqdist(c[j + 8], cc[j + 8], p[j + 8]))};
    f32x2 B[kPairs - 1], G[kPairs - 3];
#pragma unroll
    for (int j = 0; j < kPairs - 1; ++j) B[j] = A[j] + A[j + 1];
#pragma unroll
    for (int j = 0; j < kPairs - 3; ++j) G[j] = B[j] + B[j + 2];
#pragma unroll
    for (int j = 0; j < 8; ++j) C8[j] = G[j] + G[j + 4];
}

// One full evaluation's update of the wave's pixels (s2[i] = S~ of pixel i in row r8 and in row r8 + 8; rank: the candidate).
__device__ __forceinline__ void lean_update(LeanState &st, const f32x2 (&s2)[kRun], uint32_t rank) {
    // does ANY of the lane's 14 pixels pass?  (most candidates end here)
    f32x2 dm = st.thr[0] - s2[0];
    float top = __builtin_fmaxf(dm.x, dm.y);
#pragma unroll
    for (int i = 1; i < kRun; ++i) { dm = st.thr[i] - s2[i]; top = __builtin_fmaxf(top, __builtin_fmaxf(dm.x, dm.y)); }
    if (__builtin_amdgcn_readfirstlane(__ballot(top >= 0.0f) == 0ull)) return;
    const float zeroCap = __builtin_bit_cast(float, 0x00800000u + rank);       // above 0, below every non-zero S~ (>= 1)
#pragma unroll
    for (int i = 0; i < kRun; ++i) {
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
            const float s = hb ? s2[i].y : s2[i].x, thr = hb ? st.thr[i].y : st.thr[i].x;
            const bool pass = s <= thr;
            // S~ == 0: every distance is exactly 0 and so is the shader's cost: the smallest such rank wins, carried in the word
            const float cap = __builtin_fmaxf(s * kRatio, zeroCap);            // (zeroCap only for s == 0)
            const float t = pass ? __builtin_fminf(thr, cap) : thr;
            if (hb) st.thr[i].y = t; else st.thr[i].x = t;
            // a candidate that undercuts the threshold by more than the bracket is wide kills everything before it (motion_prefilter.hip,
            // `restart`); one that merely passes stands beside the best one: ambiguous
            const bool record = pass && s != 0.0f;
            const bool restart = s < thr * kRestart;
            const uint32_t bit = 1u << (16 * hb + i);
            st.amb = record ? (restart ? (st.amb & ~bit) : (st.amb | bit)) : st.amb;
            const uint32_t mask = 0xFFFFu << (16 * hb);
            st.best[i] = record ? ((st.best[i] & ~mask) | (rank << (16 * hb))) : st.best[i];
        }
    }
}

// kRankIsScan: the shaders' own tie order -- a candidate's rank IS its scan index (dy + R) * 33 + (dx + R), and phase 2 gets window
// offsets and vectors from ranks by arithmetic.  false: the intended semantics' order (lfg_set_semantics: shortest vector first,
// motion_order.hip: motion_tables) -- ranks go through rank2scan, staged in LDS once per workgroup (2.2 KB).
template <bool kRankIsScan>
__global__ __launch_bounds__(kPNT, 3) void motion_lean_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch, int W, int H,
    const uint32_t *__restrict__ order32, const uint32_t *__restrict__ rank2scan, const uint32_t *__restrict__ leanTiles, int tilesX,
    int8_t *__restrict__ mv, int mvPitch, uint32_t *__restrict__ segDone, uint32_t *__restrict__ hardTiles, uint32_t *__restrict__ hardCount,
    uint32_t *__restrict__ leanStats, int whateverTheVerdict) {
    __shared__ uint32_t sWin[kWinH * kWinW];                           // 38.2 KB packed RGBA8 search window, column-major
    __shared__ __attribute__((aligned(8))) float sSlab[kPNT / 64][kLeanSlab];
    __shared__ uint32_t sVisited[(kCand + 31) / 32];                   // the ranks the hints hold (they are not looked at twice)
    __shared__ uint32_t sLeft;                                         // some wave of the workgroup left its segment to the generic kernel
    __shared__ uint16_t sScan[kRankIsScan ? 2 : kCand + 1];            // scan index of rank r (intended order only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), seg = wave;
    // an entry: the tile, and for a rim tile above or below the interior the segments to take (bits 24..27; 0: all four) -- those
    // whose block positions all lie inside the image; the window's rows outside it are zeros, as texelFetch reads them
    const uint32_t entry = leanTiles[blockIdx.x];
    const int tile = (int)(entry & 0xFFFFFFu);
    const uint32_t segMask = (entry >> 24) != 0u ? (entry >> 24) : 0xFu;
    const bool partial = (entry >> 24) != 0u;
    const bool segOn = ((segMask >> seg) & 1u) != 0u;                  // (wave-uniform)
    const int tileY = tile / tilesX, tileX = tile - tileY * tilesX;
    const int tx0 = tileX * kPTW, ty0 = tileY * kPTH;
    const int bx0 = tx0 - kB / 2, by0 = ty0 - kB / 2;                  // image coordinates of block position (0, 0)
    // this call's order: entry = rank | window offset << 16 (motion_order.hip: motion_order_kernel); [kCand + 1]: entries in front that are hints
    // [kCand + 2]: most of the call's sample blocks have a near-exact match (motion_order_kernel) -- otherwise this is not the
    // content the kernel is for, and every workgroup leaves before it has staged anything
    // (a tile in which anything is left goes onto the list the generic kernel draws from behind its own table: here, all of them)
    if ((order32[kCand + 2] & 1u) == 0u && !whateverTheVerdict) {
        if (tid == 0 && !partial) hardTiles[atomicAdd(hardCount, 1u)] = (uint32_t)tile;      // (a rim tile's segments have their units in the plan)
        return;
    }
    if (tid == 0) sLeft = 0u;
    if (!kRankIsScan)
        for (int i = tid; i < kCand; i += kPNT) sScan[i] = (uint16_t)rank2scan[i];
    const int nHints = min((int)order32[kCand + 1], kLeanHintsMax);
    // (the hints themselves, one per lane, asked for here: in flight with the window, not a memory latency each further down)
    const uint32_t hintL = order32[lane];

    // ---- this lane's 23 block positions of the current frame (column bx0 + lane, rows by0 + 16 seg + j): all inside the image
    uint32_t c[kSegD], cc[kSegD];
    {
        // (a wave whose segment is not taken reads the rows of one that is: valid memory, never used)
        const int segC = segOn ? seg : (int)__builtin_ctz(segMask);
        const uint8_t *const column = curr + (size_t)(bx0 + min(lane, kPTW + kB - 2)) * 4u + (size_t)(by0 + kSeg * segC) * (size_t)currPitch;
#pragma unroll
        for (int j = 0; j < kSegD; ++j) c[j] = *reinterpret_cast<const uint32_t *>(column + (size_t)j * (size_t)currPitch);
    }
    // ---- the search window, prev(bx0 - R + wx, by0 - R + wy): inside prev as a whole (the host lists only such tiles), ten
    // 16-byte loads per thread in flight, then column-major into LDS
    {
        constexpr int kGroups = (kWinW + 3) / 4;                       // 24 groups of four texels per window row
        constexpr int kRounds = (kWinH * kGroups + kPNT - 1) / kPNT;   // 10
        static_assert(kPNT == 10 * kGroups + 16, "a round advances a thread by ten rows and sixteen groups");
        uint4 v[kRounds];
        int wyK = tid / kGroups, gK = tid - wyK * kGroups;
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            const int wy = min(wyK, kWinH - 1);
            if (partial) {                 // (wave-uniform) rows above or below the image: the nearest row is loaded and dropped (a mask, not a select)
                const int gy = by0 - kR + wy;
                const uint4 t = *reinterpret_cast<const uint4 *>(prev + (size_t)min(max(gy, 0), H - 1) * (size_t)prevPitch + (size_t)(bx0 - kR + 4 * gK) * 4u);
                const uint32_t keep = (gy >= 0 && gy < H) ? 0xFFFFFFFFu : 0u;
                v[k] = uint4{t.x & keep, t.y & keep, t.z & keep, t.w & keep};
            } else
            v[k] = *reinterpret_cast<const uint4 *>(prev + (size_t)(by0 - kR + wy) * (size_t)prevPitch + (size_t)(bx0 - kR + 4 * gK) * 4u);
            gK += 16; wyK += 10;
            if (gK >= kGroups) { gK -= kGroups; wyK += 1; }
        }
        int wyS = tid / kGroups, gS = tid - wyS * kGroups;
#pragma unroll
        for (int k = 0; k < kRounds; ++k) {
            if (wyS < kWinH) {
                uint32_t *dst = sWin + (4 * gS) * kWinH + wyS;
                dst[0] = v[k].x; dst[kWinH] = v[k].y; dst[2 * kWinH] = v[k].z;
                if (4 * gS + 3 < kWinW) dst[3 * kWinH] = v[k].w;
            }
            gS += 16; wyS += 10;
            if (gS >= kGroups) { gS -= kGroups; wyS += 1; }
        }
    }
#pragma unroll
    for (int j = 0; j < kSegD; ++j) cc[j] = __builtin_amdgcn_udot4(c[j], c[j], 0x4B000000u, false);     // 2^23 + |c|^2 as float bits
    if (wave == 0) {                       // the set of ranks the hints hold (one wave: its DS operations execute in order)
        if (lane < (kCand + 31) / 32) sVisited[lane] = 0u;
        wave_lds_sync();
        if (lane < nHints) atomicOr(&sVisited[(hintL & 0xFFFFu) >> 5], 1u << (hintL & 31u));
    }
    __syncthreads();                       // window and hint set in place; the waves do not meet again before the end
    if (!segOn) { __syncthreads(); return; }   // (the barrier the other waves end with)

    const lean_lds_u32 winCol = (lean_lds_u32)(sWin + min(lane, kPTW + kB - 2) * kWinH + kSeg * seg);       // this lane's position column
    const lean_lds_u32 winSeg = (lean_lds_u32)(sWin + kSeg * seg);
    const int r8 = lane & 7, q = lane >> 3;
    float *const slabW = sSlab[wave] + lane;
    const lean_lds_f32 slabR = (lean_lds_f32)(sSlab[wave]) + r8 * kLeanSlabP + kRun * q;

    // the current frame's texels at the 14 lattice points of this segment, wave-uniform: scalar registers for the whole unit
    uint32_t cLat[kLatCols][kLatRows];
#pragma unroll
    for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
        for (int t = 0; t < kLatRows; ++t) cLat[ci][t] = (uint32_t)__builtin_amdgcn_readlane((int)c[kLatR0 + 8 * t], kLatC0 + 8 * ci);
    }
    LeanState st;
#pragma unroll
    for (int i = 0; i < kRun; ++i) { st.thr[i] = f32x2{__builtin_inff(), __builtin_inff()}; st.best[i] = 0u; }
    st.amb = 0u;
    float waveThr = __builtin_inff();      // the largest threshold of the wave's pixels
    uint32_t zeroBound = 0xFFFFFFFFu;      // every pixel owns a zero-cost candidate: the largest of their ranks (only earlier ranks can matter)
    auto refresh = [&]() {
        uint32_t k = 0u;
#pragma unroll
        for (int i = 0; i < kRun; ++i) {
            const float fx = st.thr[i].x, fy = st.thr[i].y;
            k = max(k, max(__builtin_bit_cast(uint32_t, fx), __builtin_bit_cast(uint32_t, fy)));
        }
        k = wave_max_u32(k);
        zeroBound = k < 0x00800000u + (uint32_t)kCand ? k - 0x00800000u : 0xFFFFFFFFu;
        waveThr = __builtin_bit_cast(float, k);
    };
    int evals = 0;
    bool left = false;                     // the segment is left to the generic kernel

    // One candidate in full: exact-match shortcut, else distances -> column sums -> slab (two halves of eight rows) -> row sums.
    auto evaluate = [&](uint32_t ord) {
        const uint32_t rank = ord & 0xFFFFu;
        uint32_t p[kSegD];
        const lean_lds_u32 w = winCol + (ord >> 16);
#pragma unroll
        for (int j = 0; j < kSegD; ++j) p[j] = w[j];
        // a candidate that IS the motion reads the very bytes of the current frame at every block position: S~ = 0 for every pixel
        // (the OR of the XORs through an empty asm: left alone the compiler turns "!= 0" into 23 compares whose lane masks travel
        //  VALU -> SALU one by one)
        uint32_t diff = 0u;
#pragma unroll
        for (int j = 0; j < kSegD; ++j) diff |= p[j] ^ c[j];
        asm volatile("" : "+v"(diff));
        if (__builtin_amdgcn_readfirstlane((int)((__ballot(diff != 0u) & 0x7FFFFFFFFFFFFFFFull) == 0ull))) {
            const float zc = __builtin_bit_cast(float, 0x00800000u + rank);
#pragma unroll
            for (int i = 0; i < kRun; ++i) { st.thr[i].x = __builtin_fminf(st.thr[i].x, zc); st.thr[i].y = __builtin_fminf(st.thr[i].y, zc); }
            return;
        }
        ++evals;
#ifdef LFG_LEAN_DIAG_NO_EVAL             // (timing experiment, wrong vectors: the candidate counts as an exact match)
        {
            const float zc = __builtin_bit_cast(float, 0x00800000u + rank);
#pragma unroll
            for (int i = 0; i < kRun; ++i) { st.thr[i].x = __builtin_fminf(st.thr[i].x, zc); st.thr[i].y = __builtin_fminf(st.thr[i].y, zc); }
            return;
        }
#endif
        f32x2 C8[8], X[kRunIn];
        lean_column_sums(p, c, cc, C8);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 8; ++r) slabW[r * kLeanSlabP] = half ? C8[r].y : C8[r].x;
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < kRunIn; ++i) { const float t = slabR[i]; if (half) X[i].y = t; else X[i].x = t; }
        }
        wave_lds_sync();
        f32x2 h2[kRunIn - 1], h4[kRunIn - 3], s2[kRun];
#pragma unroll
        for (int i = 0; i < kRunIn - 1; ++i) h2[i] = X[i] + X[i + 1];
#pragma unroll
        for (int i = 0; i < kRunIn - 3; ++i) h4[i] = h2[i] + h2[i + 2];
#pragma unroll
        for (int i = 0; i < kRun; ++i) s2[i] = h4[i] + h4[i + 4];
        lean_update(st, s2, rank);
    };

    // The one-point test of `kA` candidates per lane at once (prefilter_tests.inc, "LOOKAHEAD"): does some lattice point of the candidate
    // come within the wave's largest threshold of the current frame's texel there?  By exact compare while the thresholds stand
    // for zero costs, by sums of absolute differences while they are small (a distance is at least half its SAD), by the
    // squared distance otherwise.
    auto latticeKeep = [&](const uint32_t (&off)[kLeanAhead], bool (&keep)[kLeanAhead]) {
        uint32_t tex[kLeanAhead][kLatCols][kLatRows];
#pragma unroll
        for (int a = 0; a < kLeanAhead; ++a) {
            const lean_lds_u32 w = winSeg + off[a];
#pragma unroll
            for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                for (int t = 0; t < kLatRows; ++t) tex[a][ci][t] = w[(kLatC0 + 8 * ci) * kWinH + kLatR0 + 8 * t];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // (three loops, not one with the choice inside: the choice is wave-uniform, and inside the loop it was three scalar
        //  branches per lattice point)
        if (waveThr < 0.5f) {
            uint32_t acc[kLeanAhead];                                              // the smallest XOR is 0: VALU only
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) acc[a] = 0xFFFFFFFFu;
#pragma unroll
            for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                for (int t = 0; t < kLatRows; ++t) {
#pragma unroll
                    for (int a = 0; a < kLeanAhead; ++a) acc[a] = min(acc[a], tex[a][ci][t] ^ cLat[ci][t]);
                }
            }
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) keep[a] = acc[a] == 0u;
        } else if (waveThr < kSadTestMax) {
            uint32_t acc[kLeanAhead];
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) acc[a] = 0xFFFFFFFFu;
#pragma unroll
            for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                for (int t = 0; t < kLatRows; ++t) {
#pragma unroll
                    for (int a = 0; a < kLeanAhead; ++a) acc[a] = min(acc[a], __builtin_amdgcn_sad_u8(cLat[ci][t], tex[a][ci][t], 0u));
                }
            }
            const float sadMax = (2.0f * waveThr) * 1.00001f;                       // SAD > 2 thr => distance > thr
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) keep[a] = (float)acc[a] <= sadMax;
        } else {
            uint32_t acc[kLeanAhead];
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) acc[a] = 0x7F800000u;
#pragma unroll
            for (int ci = 0; ci < kLatCols; ++ci) {
#pragma unroll
                for (int t = 0; t < kLatRows; ++t) {
                    const uint32_t ccT = __builtin_amdgcn_udot4(cLat[ci][t], cLat[ci][t], 0x4B000000u, false);
#pragma unroll
                    for (int a = 0; a < kLeanAhead; ++a) acc[a] = min(acc[a], __builtin_bit_cast(uint32_t, lean_sqdist(cLat[ci][t], ccT, tex[a][ci][t])));
                }
            }
            // n > thr^2 (1 + 2^-20) => sqrt(n) exceeds thr beyond v_sqrt_f32's ulp
            const uint32_t thrSqBits = __builtin_bit_cast(uint32_t, (waveThr * waveThr) * 1.000001f);
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) keep[a] = !(acc[a] > thrSqBits);
        }
    };

    // ---- ONE loop over three kinds of steps, so that the evaluation above exists once in the code:
    //   phase 0  the call's hints one by one while the wave has no small threshold yet (the first alone: under a pan it is all a
    //            wave ever evaluates in full);
    //   phase 1  the remaining hints as one batch, one per lane, against the thresholds the first ones left;
    //   phase 2  every other candidate by rank (= scan index: rank r is dx = r % 33 - 16, dy = r / 33 - 16), 192 per pass.
    // A step yields up to three masks of lanes whose candidate has to be evaluated in full; after them the thresholds are reduced.
    int phase = 0, h = 0, r0 = 0;
    for (;;) {
        unsigned long long m[kLeanAhead];
        uint32_t ordA[kLeanAhead];
#pragma unroll
        for (int a = 0; a < kLeanAhead; ++a) { m[a] = 0ull; ordA[a] = 0u; }
        if (phase == 0) {
            if (h >= nHints || (h > 0 && waveThr < kOnePointOnly) || evals > LFG_LEAN_EVALS_MAX) { phase = 1; continue; }
            ordA[0] = (uint32_t)__builtin_amdgcn_readlane((int)hintL, h);
            m[0] = 1ull;
            ++h;
        } else if (phase == 1) {
            phase = 2;
            left = !(waveThr < kOnePointOnly);         // not the easy kind: the other tests, the narrow search, the hand-over are the generic kernel's
            if (left) break;
            if (h >= nHints) continue;
            ordA[0] = hintL;                            // lane l takes hint l
            uint32_t off[kLeanAhead];
            bool keep[kLeanAhead];
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) off[a] = hintL >> 16;      // (the one test, several times over: once per unit)
            latticeKeep(off, keep);
            m[0] = __ballot(keep[0] && lane >= h && lane < nHints && (hintL & 0xFFFFu) < zeroBound);
            h = nHints;
            if (m[0] == 0ull) continue;
        } else {
            const int bound = (int)min(zeroBound, (uint32_t)kCand);
            if (r0 >= bound) break;
#ifdef LFG_LEAN_DIAG_NO_RANKS            // (timing experiment, wrong vectors possible: no pass over the ranks)
            break;
#endif
            uint32_t off[kLeanAhead];
            bool need[kLeanAhead], keep[kLeanAhead];
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) {
                const int r = r0 + 64 * a + lane, rc = min(r, kCand - 1);
                const uint32_t sc = kRankIsScan ? (uint32_t)rc : (uint32_t)sScan[rc];
                const uint32_t dyi = (sc * 1986u) >> 16, dxi = sc - 33u * dyi;     // sc / 33, sc % 33 for sc < 1089
                off[a] = dxi * (uint32_t)kWinH + dyi;
                ordA[a] = (uint32_t)rc | (off[a] << 16);
                need[a] = r < bound && ((sVisited[rc >> 5] >> (rc & 31)) & 1u) == 0u;
            }
            latticeKeep(off, keep);
            unsigned long long anyM = 0ull;
#pragma unroll
            for (int a = 0; a < kLeanAhead; ++a) { m[a] = __ballot(need[a] && keep[a]); anyM |= m[a]; }
            r0 += 64 * kLeanAhead;
            if (anyM == 0ull) continue;
        }
        // the survivors, one after the other (slot by slot; the slot's index is wave-uniform, the selects are scalar)
        for (;;) {
            int a = -1;
#pragma unroll
            for (int k = kLeanAhead - 1; k >= 0; --k) a = m[k] != 0ull ? k : a;
            if (a < 0) break;
            unsigned long long ma = 0ull;
#pragma unroll
            for (int k = 0; k < kLeanAhead; ++k) ma = a == k ? m[k] : ma;
            const int b = __builtin_ctzll(ma);
            uint32_t ord = 0u;
#pragma unroll
            for (int k = 0; k < kLeanAhead; ++k) {
                const uint32_t ok = (uint32_t)__builtin_amdgcn_readlane((int)ordA[k], b);
                ord = a == k ? ok : ord;
                m[k] = a == k ? (m[k] & (m[k] - 1ull)) : m[k];
            }
            if ((ord & 0xFFFFu) < zeroBound) evaluate(ord);
        }
        refresh();
        if (phase == 2 && (evals > LFG_LEAN_EVALS_MAX || !(waveThr < kOnePointOnly))) { left = true; break; }
    }

    // ---- the answers: the rank in a zero-cost word, or the single best candidate; anything ambiguous leaves the segment
    uint32_t rankOf[2][kRun];
    bool open = left;
#pragma unroll
    for (int i = 0; i < kRun; ++i) {
        const float fx = st.thr[i].x, fy = st.thr[i].y;
        rankOf[0][i] = fx < 0.5f ? __builtin_bit_cast(uint32_t, fx) - 0x00800000u : (st.best[i] & 0xFFFFu);
        rankOf[1][i] = fy < 0.5f ? __builtin_bit_cast(uint32_t, fy) - 0x00800000u : (st.best[i] >> 16);
        open = open || (!(fx < 0.5f) && ((st.amb >> i) & 1u)) || (!(fy < 0.5f) && ((st.amb >> (16 + i)) & 1u));
    }
    if (__builtin_amdgcn_readfirstlane(__ballot(open) == 0ull)) {
        // the 16 x 56 vectors through the wave's slab, out as 112 contiguous bytes per image row (prefilter_epilogue.inc)
        uint16_t *const rows = reinterpret_cast<uint16_t *>(sSlab[wave]);
        wave_lds_sync();
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) {
#pragma unroll
            for (int i = 0; i < kRun; ++i) {
                const uint32_t r = kRankIsScan ? rankOf[hb][i] : (uint32_t)sScan[min(rankOf[hb][i], (uint32_t)kCand - 1u)];
                const uint32_t dyi = (r * 1986u) >> 16, dxi = r - 33u * dyi;
                rows[(8 * hb + r8) * kPTW + kRun * q + i] = (uint16_t)(uint8_t)(int8_t)((int)dxi - kR) | (uint16_t)((uint16_t)(uint8_t)(int8_t)((int)dyi - kR) << 8);
            }
        }
        wave_lds_sync();
        const int half = lane / 28, d = lane - half * 28;              // 28 dwords = one row of the tile
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int row = 2 * rr + half;
            if (lane < 56)
                *reinterpret_cast<uint32_t *>(mv + (size_t)(ty0 + kSeg * seg + row) * (size_t)mvPitch + (size_t)(tx0 + 2 * d) * 2u) =
                    *reinterpret_cast<const uint32_t *>(rows + row * kPTW + 2 * d);
        }
        if (lane == 0) segDone[tile * (kPTH / kSeg) + seg] = 1u;
    }
    if (__builtin_amdgcn_readfirstlane(__ballot(open) != 0ull) && lane == 0) sLeft = 1u;
    __syncthreads();
    if (tid == 0 && sLeft != 0u && !partial) hardTiles[atomicAdd(hardCount, 1u)] = (uint32_t)tile;
#ifdef LFG_LEAN_STATS                       // (diagnostic build only: 8,576 device-scope atomics on two words are 120 us of a 70 us launch)
    if (leanStats && lane == 0) atomicAdd(&leanStats[__builtin_amdgcn_readfirstlane(__ballot(open) == 0ull) ? 0 : 1], 1u);
#endif
}

// The tiles the lean kernel may take, from the plan's whole-tile units: window inside prev in the staging loop's terms (groups of
// four texels, 96 wide), every block position inside the image.
bool lean_tile_ok(int tile, int tilesX, int W, int H) {
    const int ty = tile / tilesX, tx = tile - ty * tilesX;
    const int bx0 = tx * kPTW - kB / 2, by0 = ty * kPTH - kB / 2;
    return bx0 - kR >= 0 && bx0 - kR + 96 <= W && by0 - kR >= 0 && by0 - kR + kWinH <= H &&
           tx * kPTW + kPTW <= W && ty * kPTH + kPTH <= H && bx0 + kPTW + kB - 2 + kR < W && by0 + kPTH + kB - 2 + kR < H;
}

// A segment of a tile that is not the kernel's as a whole: the tile's columns as above, and the segment's rows -- its 23 rows of
// block positions and its 16 rows of pixels inside the image, and no candidate's block outside prev altogether (no plateaus).
bool lean_segment_ok(int tile, int seg, int tilesX, int W, int H) {
    const int ty = tile / tilesX, tx = tile - ty * tilesX;
    const int bx0 = tx * kPTW - kB / 2, by0 = ty * kPTH - kB / 2, y0 = ty * kPTH + kSeg * seg;
    return bx0 - kR >= 0 && bx0 - kR + 96 <= W && tx * kPTW + kPTW <= W && bx0 + kPTW + kB - 2 + kR < W &&
           by0 + kSeg * seg >= 0 && by0 + kSeg * seg + kSegD - 1 < H && y0 + kSeg <= H &&
           y0 + kB / 2 - 1 - kR >= 0 && y0 + kSeg - 1 - kB / 2 + kR < H;
}

bool lean_frames_ok(const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv) {
    return (curr.width & 3u) == 0u && (prev.pitch & 15u) == 0u && ((uintptr_t)prev.data & 15u) == 0u &&
           (curr.pitch & 3u) == 0u && (mv.pitch & 3u) == 0u && ((uintptr_t)mv.data & 3u) == 0u;
}

hipError_t launch_motion_lean(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                              const uint32_t *order32, const uint32_t *rank2scan, bool rankIsScan, const uint32_t *leanTiles, int nTiles, int tilesX,
                              uint32_t *segDone, uint32_t *hardTiles, uint32_t *hardCount, uint32_t *stats, bool whateverTheVerdict) {
    if (nTiles <= 0) return hipSuccess;
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)nTiles), dim3(kPNT), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch, (int)curr.width, (int)curr.height,
                           order32, rank2scan, leanTiles, tilesX, (int8_t *)mv.data, (int)mv.pitch, segDone, hardTiles, hardCount, stats, whateverTheVerdict ? 1 : 0);
    };
    if (rankIsScan) launch(motion_lean_kernel<true>); else launch(motion_lean_kernel<false>);
    return hipGetLastError();
}

}  // namespace lfg
