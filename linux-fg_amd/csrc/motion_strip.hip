// motion_strip.hip -- the strips a pan exposes (shaders/motion.comp:27-52 for pixels that have NO match), as a kernel of their own.
// OPT-IN (LFG_MOTION_STRIP=1): built to VERDICT r4's item 1, exact, tested -- and measured SLOWER than the path it replaces.  What the
// round learnt from it is at the end of this comment.
//
// A frame that moves as a whole exposes new content along one or two edges: a dozen pixel columns on the left of the 4K benchmark
// frame, eight rows at its bottom.  No candidate matches there, no partial-distortion test can drop anything, and every one of the
// 1,089 candidates has to be summed in full for every such pixel.  The persistent kernel does that inside its work unit (narrow
// search, row band: prefilter_narrow.inc, prefilter_rowband.inc) at 256 VGPRs and two waves per SIMD: 262 units of 225 - 232 us.
// This kernel is that search written for what it is -- EVERY candidate, a few pixels -- and nothing else:
//
//   * one work item = 57 pixels ALONG an edge x the band ACROSS it (kStripCols columns, or kStripRows rows) x all candidates; a
//     workgroup of sixteen waves, wave w taking the ranks = w mod 16;
//   * lane = block position along the edge (64 positions serve 57 pixels), so all 64 lanes compute distances for every candidate:
//     band + 7 distances along the in-lane axis (those that fall outside the image are known when the code is written and never
//     computed), an in-lane pairwise tree to the band's 8-sums, and the 8-sum ACROSS lanes: lane l + 1 by DPP (v_add_f32_dpp
//     wave_shl:1), lanes l + 2 and l + 4 through the LDS crossbar (ds_bpermute_b32) -- no transposition, no slab; S~ is again a
//     depth-6 pairwise tree over exactly rounded integer squared distances and v_sqrt_f32: within 9 u of 255 x the real-arithmetic
//     cost (motion_prefilter.hip, "Bracket");
//   * per pixel ONE register: key = (12 mantissa bits of S~ | rank), kept as the running minimum by v_min_u32; a second instruction
//     per pixel (v_sad_u32) measures how close the candidate came to the holder, and only when some pixel of the wave saw a near
//     miss or a near undercut (|key - holder| <= kNearD: a superset of "within the bracket") does a scalar branch record a key in
//     the pixel's short list.  Every candidate within the bracket of the final minimum is then either a wave's final holder or
//     recorded (proof below), and the shader's own 64-term chain decides among those few;
//   * 95 VGPRs, 53 KB of LDS, 1,024 threads: one workgroup per CU.
//
// The kernel decides its pixels COMPLETELY -- vectors written, no lists, no resolve kernel -- and says which pixels it took in
// two small tables (colBand[y]: the left / right band of row y; rowBand[x]: the top / bottom band of column x) that the persistent
// kernel's rim units and the resolve kernel read: those pixels are parked there like pixels outside the image.  Which strips exist
// is read off the call's top hint (a vector with dx < 0 exposes the left edge, dy > 0 the bottom ...): a wrong guess costs time,
// never a result -- pixels the persistent kernel does not find in the tables it searches itself, as before.
//
// Exactness.  K(m) = (T(m) | rank), T = S~ truncated to 12 mantissa bits, so T <= S~ < T (1 + 2^-12).  Let C be the candidates with
// S~(m) <= kRatio min S~ -- the only ones that can be the shader's first strict minimum ("Bracket").  For m in C and b the holder of
// the smallest key: T(m) <= kRatio min S~ < kRatio (1 + 2^-12) T(b), i.e. less than 2.7 units of T's last place above T(b) whatever
// the binade: K(m) - K(b) <= 2 * 2048 + 2047 < kNearD.  x is NEAR when |K(x) - K(holder)| <= kNearD; a near MISS records x, a near
// UNDERCUT records the holder that x replaces (if that holder is the wave's own candidate: a holder adopted from the pool is
// recorded by the wave that owns it).  Take m in C, m != b, in b's wave: visited after b, it is near b; visited before, it is either
// near the holder of that moment (recorded), or more than kNearD above it (then above K(b) + kNearD: not in C), or becomes the
// holder unrecorded -- but then every later change of holder on the way down to b happens within kNearD (all those keys lie
// between K(b) and K(m)) and records the holder it replaces, m first; a holder that loses its key to a POOLED minimum (the waves
// publish and adopt their minima every kStripSync candidates, without a barrier) is compared with it like a candidate (`adopt`).
// Candidates of C in another wave: the same argument with that wave's own sequence of holders.  The first kStripWarm candidates of
// a wave run without recording (the chance of a near event falls like 1 / n with the n candidates a minimum has seen: from nothing,
// lists of six overflowed on a tenth of the pixels) and are visited once more at the end.  S~ = 0 means cost exactly 0: the key IS
// the rank, the smallest rank wins, no chain.  Candidates whose block leaves prev altogether tie exactly (a plateau,
// motion_prefilter.hip): across the edge only the first member in tie order keeps its key (`rep`), along it the recording path
// skips pairs of members, and the chain phase maps every candidate to the first member of its plateau as the resolve kernel does.
//
// MEASURED (round 5, MI355X, the benchmark's 4K pan; NOTES_r05.md section 3).  The full search costs 2.5 - 3 SIMD-cycles per
// (pixel, candidate) HERE and in the persistent kernel's narrow search alike -- 33 cycles of VALU per distance (two v_dot4 at 7.3,
// v_sqrt_f32 at 10, two adds: tools/bench_dpp.hip), 1.4 - 2 distances per pixel and candidate, then trees, cross-lane sums and the
// key -- and both are bound by that arithmetic, not by latency: sixteen waves per CU at 95 VGPRs buy nothing over eight at 256.
// The kernel takes 258 us for the pan's 106 items (one CU each, 150 CUs idle beside it) and shortens the persistent kernel's rim
// units from 110 to 57 us on average; together: 3,764 frames/s against 3,920 without it (three frames in flight), 1,790 against
// 2,710 one frame at a time.  With the bands of all four edges (the nearly matching ones too) the rim units fall to 35 us and the
// kernel doubles.  Smaller workgroups, bands of 8 / 12 / 16 columns, DPP against ds_bpermute: all measured, none changes the sign.
#include "lfg_motion_common.hpp"

#include <algorithm>
#include <vector>

namespace lfg {

constexpr int kStripPix = 64 - (kB - 1);            // 57 pixels along the lane axis per item
constexpr int kStripThreads = 1024, kStripWaves = kStripThreads / 64;   // sixteen waves: one workgroup fills a CU (four waves a SIMD at <= 128 VGPRs)
constexpr int kStripWinLane = 64 + 2 * kR;          // 96 window positions along the lane axis
constexpr int kStripWarm = 4;                       // candidates a wave evaluates before the waves pool their minima (and once more at the end)
constexpr int kStripSync = 16;                      // ... and every so many candidates a wave publishes its minima and adopts the pooled ones
constexpr int kStripListCap = 6;                    // recorded keys per pixel; more: the tile goes through the literal kernel
constexpr uint32_t kKeyMask = 0xFFFFF800u;          // 12 mantissa bits of S~ above an 11-bit rank
constexpr uint32_t kNearD = 3u * 2048u + 2047u;     // see "Exactness"
constexpr int kStripSlots = 64 * kStripCols;        // pixel slots of an item in LDS (lane * pixels across + i), for the wider kind of item

typedef const __attribute__((address_space(3))) uint32_t *strip_lds_u32;

// lane l <- lane l + 1 (the last lane reads 0): v_mov_b32_dpp / folded into the add that uses it
// (mov_dpp, not update_dpp: with every row and bank enabled and bound_ctrl set no lane keeps its old value, and an `old`
//  operand of 0 cost a v_mov_b32 in front of each of the 64 shifts of a candidate)
__device__ __forceinline__ float strip_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
// |a - b| of two unsigned words in one instruction (hipcc has builtins for the byte and halfword SADs only)
__device__ __forceinline__ uint32_t strip_absdiff(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// One key into a pixel's list.  A full list gives up an entry that is STALE -- more than kNearD above `floor`, a minimum of this pixel
// that some wave holds now: the final minimum is no larger, the entry cannot lie within its bracket -- and only when there is none
// does the pixel count as overflowed (bit 31 of the count: its tile goes through the literal kernel).  The eight waves share the
// list: the slot is taken by compare-and-swap.
__device__ __forceinline__ void strip_record(uint32_t *cnt, uint32_t *list, uint32_t key, uint32_t floor) {
    const uint32_t at = atomicAdd(cnt, 1u) & 0x7FFFFFFFu;
    if (at < (uint32_t)kStripListCap) { list[at] = key; return; }
    for (int k = 0; k < kStripListCap; ++k) {        // (a recorded key far BELOW this one: another wave's minimum is lower, this key is the stale one)
        const uint32_t e = list[k];
        if (key > e && key - e > kNearD) return;
    }
    for (int k = 0; k < kStripListCap; ++k) {
        const uint32_t e = list[k];
        if (e > floor && e - floor > kNearD && atomicCAS(&list[k], e, key) == e) return;
    }
    atomicOr(cnt, 0x80000000u);
}
// (a & m) | r in one instruction
__device__ __forceinline__ uint32_t strip_and_or(uint32_t a, uint32_t m, uint32_t r) {
    uint32_t d;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "s"(r));      // (one scalar operand per instruction on this chip: the mask rides in a register)
    return d;
}
// sum over lanes l .. l + 7 as a pairwise tree (lanes 57 .. 63 end up with partial sums nobody reads)
__device__ __forceinline__ float strip_sum8_lanes(float v) {
    const float t2 = v + strip_shl1(v);
    const float t4 = t2 + strip_shl1(strip_shl1(t2));
    const float t8 = t4 + strip_shl1(strip_shl1(strip_shl1(strip_shl1(t4))));
    return t8;
}

struct StripArgs {
    const uint8_t *prev, *curr;
    int prevPitch, currPitch, W, H;
    int8_t *mv; int mvPitch;
    const uint32_t *order32, *rank2scan;
    uint32_t *colBand, *rowBand;                 // [H], [W]: bit 0 = low band decided, bit 1 = high band (lfg_motion_tile.hpp) -- cleared with the call's control area
    uint32_t *tileFlags, *flagged; int flagTilesX;   // the literal kernel's tile flags, count and list (motion_prefilter.hip: gaveUp)
    int chunksTall, chunksWide;                  // items per side
};

// One item.  kTall: the band is kOut pixel COLUMNS (lane axis = y, in-lane axis = x); else kOut pixel ROWS (lane axis = x).
// kEdge: 0 = the item's pixels start at the image's low edge (I0 = 0: the four in-lane positions in front of them are outside the image),
// 1 = they end at its high edge (I0 = size - kOut: the last three are), 2 = neither (the inner half of a band of columns) -- known when
// the code is written, so those distances are never computed.  `bit`: the item's bit in the table of decided pixels.
template <bool kTall, int kOut, int kEdge, bool kRankIsScan>
__device__ __forceinline__ void strip_item(const StripArgs &a, const int L0, const int I0, const uint32_t bit, uint32_t *sWin, uint32_t *sGmin, uint32_t *sCnt, uint32_t *sList,
                                           uint16_t *sFinal, uint16_t *sHard, float (*sDist)[kB * kB], const uint16_t *sScan, uint32_t &sHardN) {
    constexpr int kIn = kOut + kB - 1;               // block positions per lane
    constexpr int kWinIn = kIn + 2 * kR;             // window extent along the in-lane axis
    constexpr int kP = kWinIn | 1;                   // window pitch (words): odd, so the lanes of a read fall into distinct banks
    static_assert(kOut <= kStripCols, "the LDS arrays are laid out for kStripCols pixels across the edge at most");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.W, H = a.H;
    const int laneSize = kTall ? H : W;              // image extent along the lane axis
    // image coordinates of (lane-axis position u, in-lane position v) of the WINDOW: L0 - 4 - R + u, I0 - 4 - R + v
    const int wl0 = L0 - kB / 2 - kR, wi0 = I0 - kB / 2 - kR;

    // ---- this lane's block positions of the current frame (lane-axis coordinate L0 - 4 + lane).  In-lane positions jLo .. jHi - 1
    // lie inside the image (the others never do: the band sits at the edge); a LANE whose coordinate lies outside it (the first
    // and the last item of a side only) contributes nothing: its sums are zeroed further down.
    constexpr int jLo = kEdge == 0 ? kB / 2 : 0, jHi = kEdge == 1 ? kIn - (kB / 2 - 1) : kIn;
    auto inImage = [](int j) constexpr { return j >= jLo && j < jHi; };
    uint32_t c[kIn], cc[kIn];
    const int gl = L0 - kB / 2 + lane;
    const bool laneIn = gl >= 0 && gl < laneSize;
    const int edgeItem = __builtin_amdgcn_readfirstlane((int)(L0 - kB / 2 < 0 || L0 - kB / 2 + 63 >= laneSize));
    {
#pragma unroll
        for (int j = jLo; j < jHi; ++j) {
            const int gi = I0 - kB / 2 + j;
            const int x = kTall ? gi : gl, y = kTall ? gl : gi;
            c[j] = *reinterpret_cast<const uint32_t *>(a.curr + (size_t)clampi(y, 0, H - 1) * (size_t)a.currPitch + (size_t)clampi(x, 0, W - 1) * 4u);
        }
    }
    // ---- the search window of prev, zero outside the image (texelFetch), into LDS as sWin[u * kP + v]
    for (int e = tid; e < kStripWinLane * kWinIn; e += kStripThreads) {
        // tall: v runs along x -- consecutive threads read consecutive texels of an image row; wide: u runs along x
        const int u = kTall ? e / kWinIn : e % kStripWinLane, v = kTall ? e % kWinIn : e / kStripWinLane;
        const int x = kTall ? wi0 + v : wl0 + u, y = kTall ? wl0 + u : wi0 + v;
        const uint32_t t = *reinterpret_cast<const uint32_t *>(a.prev + (size_t)clampi(y, 0, H - 1) * (size_t)a.prevPitch + (size_t)clampi(x, 0, W - 1) * 4u);
        sWin[u * kP + v] = (x >= 0 && x < W && y >= 0 && y < H) ? t : 0u;
    }
    for (int e = tid; e < kStripSlots; e += kStripThreads) { sGmin[e] = 0xFFFFFFFFu; sCnt[e] = 0u; }
    if (tid == 0) sHardN = 0u;
#pragma unroll
    for (int j = jLo; j < jHi; ++j) {
        cc[j] = __builtin_amdgcn_udot4(c[j], c[j], 0x4B000000u, false);     // 2^23 + |c|^2 as float bits
        asm volatile("" : "+v"(cc[j]));    // (opaque: left alone the compiler recomputes all of them for every candidate rather than keep them)
    }
    __syncthreads();

    // this lane's pixels: lane-axis coordinate L0 + lane (lanes 0 .. 56), in-lane coordinates I0 .. I0 + kOut - 1
    const bool laneOk = lane < kStripPix && L0 + lane < laneSize;
    uint32_t mn[kOut];
#pragma unroll
    for (int i = 0; i < kOut; ++i) mn[i] = 0xFFFFFFFFu;
    const strip_lds_u32 winLane = (strip_lds_u32)sWin + lane * kP;
    auto leaves = [&](int px, int py, uint32_t rank) {
        const uint32_t sc = kRankIsScan ? rank : (uint32_t)sScan[min(rank, (uint32_t)kCand - 1u)];
        const int dyi = (int)((sc * 1986u) >> 16), dxi = (int)sc - 33 * dyi;
        return block_leaves_prev(px, py, dxi - kR, dyi - kR, W, H);
    };
    // Candidates whose block leaves prev ACROSS the edge the band lies at tie exactly for a pixel (a plateau: every texel they read is
    // an out-of-image zero) -- for the top rows of a frame that is the first 400 candidates of the scan -- and only the first of them in
    // tie order can win.  Whether a candidate is such a member depends on its in-lane displacement and the pixel's in-lane index alone,
    // the same for every lane: rep[i] = the first member for pixel slot i (kCand: none), found once per item; every other member's key
    // is set to "never" for that slot, instead of tying with the holder candidate after candidate.
    // (Plateaus ALONG the edge -- the first and the last item of a side -- stay with the recording path's own test.)
    auto inPlateau = [&](int i, uint32_t sc) {         // sc: scan index
        const int dyi = (int)((sc * 1986u) >> 16), dxi = (int)sc - 33 * dyi, dI = (kTall ? dxi : dyi) - kR;
        const int q = I0 + i;                          // the pixel's in-lane coordinate: its block reads prev at q - 4 + dI .. q + 3 + dI
        return q + kB / 2 - 1 + dI < 0 || q - kB / 2 + dI >= (kTall ? W : H);
    };
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < kOut; ++i) {
            uint32_t first = (uint32_t)kCand;
            for (int r0 = 0; r0 < kCand && first == (uint32_t)kCand; r0 += 64) {
                const int r = r0 + lane;
                const unsigned long long hit = __ballot(r < kCand && inPlateau(i, kRankIsScan ? (uint32_t)r : (uint32_t)sScan[min(r, kCand - 1)]));
                if (hit != 0ull) first = (uint32_t)(r0 + __builtin_ctzll(hit));
            }
            if (lane == 0) sHard[i] = (uint16_t)first;       // (sHard is free until the chain phase)
        }
    }
    __syncthreads();
    uint32_t rep[kOut];
#pragma unroll
    for (int i = 0; i < kOut; ++i) rep[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)sHard[i]);
    __syncthreads();
    // a holder that loses its key to a pooled, smaller one without having been compared with it: recorded if it is this wave's own and
    // within the window (see "Exactness"; the recording path's rules)
    auto adopt = [&](int i, uint32_t g) {
        if (mn[i] != g && mn[i] - g <= kNearD && !(mn[i] < 2048u && g < 2048u) && (mn[i] & 0x7FFu) % (uint32_t)kStripWaves == (uint32_t)wave) {
            const int px = kTall ? I0 + i : L0 + lane, py = kTall ? L0 + lane : I0 + i;
            const bool plateau = (mn[i] >> 11) == (g >> 11) && leaves(px, py, mn[i] & 0x7FFu) && leaves(px, py, g & 0x7FFu);
            if (!plateau) strip_record(sCnt + lane * kOut + i, sList + (lane * kOut + i) * kStripListCap, mn[i], g);
        }
        mn[i] = g;
    };

    // ---- every candidate of this wave's share.  A near event is measured against the RUNNING minimum, and running minima fall
    // fast at first: the chance that candidate n of a search comes within the window of the minimum so far goes like 1 / n, so a
    // wave that starts from nothing records ~0.5 events per pixel over its 136 candidates (first version: lists of six overflowed
    // on a tenth of the pixels and tiles went through the literal kernel by the dozen).  So the first kStripWarm candidates of
    // every wave run WITHOUT recording, the eight waves pool their minima (64 candidates seen: every wave continues from that
    // holder), the rest runs with recording -- 0.04 events per pixel and wave -- and the warm-up candidates are visited once
    // more at the end, recording this time (a key's minimum is idempotent): +6 % evaluations.
    const int lanePlus2 = ((lane + 2) & 63) * 4, lanePlus4 = ((lane + 4) & 63) * 4;       // ds_bpermute addresses (lanes 57 .. 63 read garbage nobody uses)
    const int nMine = (kCand - wave + kStripWaves - 1) / kStripWaves;       // ranks wave, wave + 8, ...
    for (int t = 0; t < nMine + kStripWarm; ++t) {
        if (t == kStripWarm) {             // (every wave passes here: nMine > kStripWarm)
            if (laneOk) {
#pragma unroll
                for (int i = 0; i < kOut; ++i) atomicMin(&sGmin[lane * kOut + i], mn[i]);
            }
            __syncthreads();
            if (laneOk) {
#pragma unroll
                for (int i = 0; i < kOut; ++i) mn[i] = sGmin[lane * kOut + i];
            }
        }
        else if (t > kStripWarm && ((t - kStripWarm) & (kStripSync - 1)) == 0) {
            // (no barrier: whatever the other waves have published by now; a wave's own atomics and reads stay in order)
            if (laneOk) {
#pragma unroll
                for (int i = 0; i < kOut; ++i) atomicMin(&sGmin[lane * kOut + i], mn[i]);
#pragma unroll
                for (int i = 0; i < kOut; ++i) adopt(i, sGmin[lane * kOut + i]);
            }
        }
        const bool recording = t >= kStripWarm;
        const int r = wave + kStripWaves * (t < nMine ? t : t - nMine);
        const uint32_t sc = kRankIsScan ? (uint32_t)r : (uint32_t)sScan[r];
        const uint32_t dyi = (sc * 1986u) >> 16, dxi = sc - 33u * dyi;        // sc / 33, sc % 33 for sc < 1089
        const uint32_t off = (kTall ? dyi : dxi) * (uint32_t)kP + (kTall ? dxi : dyi);
        const strip_lds_u32 w = winLane + off;
        uint32_t p[kIn];
#pragma unroll
        for (int j = jLo; j < jHi; ++j) p[j] = w[j];
        // distances: n = |c|^2 + |p|^2 - 2 c.p by dot products onto float bit patterns (motion_prefilter.hip: columnSums)
        float d[kIn];
#pragma unroll
        for (int j = jLo; j < jHi; ++j) {
            const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(p[j], p[j], cc[j], false));
            const float f2 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(c[j], p[j], 0x4B800000u, false));
            d[j] = __builtin_amdgcn_sqrtf((f1 - f2) + 8388608.0f);
        }
        // the in-lane 8-sums as a pairwise tree; a position outside the image is skipped by the shader and is simply not there
        // (sum2's flags are constants once the loops are unrolled: no instruction is spent on a term that does not exist)
        auto sum2 = [](float x, bool xIn, float y, bool yIn) { return xIn && yIn ? x + y : xIn ? x : yIn ? y : 0.0f; };
        auto in2 = [&](int j) constexpr { return inImage(j) || inImage(j + 1); };
        auto in4 = [&](int j) constexpr { return in2(j) || in2(j + 2); };
        float a2[kIn - 1], a4[kIn - 3], h8[kOut];
#pragma unroll
        for (int j = 0; j < kIn - 1; ++j) a2[j] = sum2(inImage(j) ? d[j] : 0.0f, inImage(j), inImage(j + 1) ? d[j + 1] : 0.0f, inImage(j + 1));
#pragma unroll
        for (int j = 0; j < kIn - 3; ++j) a4[j] = sum2(a2[j], in2(j), a2[j + 2], in2(j + 2));
#pragma unroll
        for (int i = 0; i < kOut; ++i) h8[i] = sum2(a4[i], in4(i), a4[i + 4], in4(i + 4));
        if (edgeItem) {                    // (wave-uniform) a lane outside the image: none of its positions exists
            asm volatile("; lanes outside the image (strip)");
#pragma unroll
            for (int i = 0; i < kOut; ++i) h8[i] = laneIn ? h8[i] : 0.0f;
        }
        // S~ of pixel (lane, i): the 8-sum across lanes, level by level over all kOut values -- lane l adds lane l + 1 in one DPP
        // instruction (wave_shl:1); lanes l + 2 and l + 4 come through the LDS crossbar (ds_bpermute_b32: no memory, the lanes' words
        // permuted), whose latency the kOut independent values hide.  Wavefront shifts go one lane at a time and cost the VALU twice
        // a plain instruction (tools/bench_dpp.hip): all three levels by DPP were 7 instructions per value and a third of a candidate.
        float t2[kOut], t4[kOut], sum[kOut];
#pragma unroll
        for (int i = 0; i < kOut; ++i) t2[i] = h8[i] + strip_shl1(h8[i]);
#pragma unroll
        for (int i = 0; i < kOut; ++i) t4[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(lanePlus2, __builtin_bit_cast(int, t2[i])));
#pragma unroll
        for (int i = 0; i < kOut; ++i) t4[i] = t2[i] + t4[i];
#pragma unroll
        for (int i = 0; i < kOut; ++i) sum[i] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(lanePlus4, __builtin_bit_cast(int, t4[i])));
        // ... its key, and how near the key came to the holder's
        uint32_t key[kOut], nearMin = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < kOut; ++i) {
            const float s = t4[i] + sum[i];
            key[i] = strip_and_or(__builtin_bit_cast(uint32_t, s), kKeyMask, (uint32_t)r) | ((inPlateau(i, sc) && (uint32_t)r != rep[i]) ? 0xFFFFFFFFu : 0u);
            nearMin = min(nearMin, strip_absdiff(key[i], mn[i]));
        }
        nearMin = laneOk ? nearMin : 0xFFFFFFFFu;
        if (recording && __builtin_amdgcn_readfirstlane((int)(__ballot(nearMin <= kNearD) != 0ull))) {
            asm volatile("; near event (strip)");
#pragma unroll
            for (int i = 0; i < kOut; ++i) {
                const bool near = laneOk && strip_absdiff(key[i], mn[i]) <= kNearD;
                if (__builtin_amdgcn_readfirstlane((int)(__ballot(near) != 0ull))) {
                    if (near) {
                        const int px = kTall ? I0 + i : L0 + lane, py = kTall ? L0 + lane : I0 + i;
                        const uint32_t holder = mn[i];
                        // two zero-cost candidates: the smaller rank holds the key, nothing to decide; two members of one
                        // plateau (same truncated S~, both blocks outside prev): likewise
                        bool skip = (key[i] < 2048u && holder < 2048u) || key[i] == holder;       // (== : a warm-up candidate that still holds the key)
                        if (!skip && (key[i] >> 11) == (holder >> 11)) skip = leaves(px, py, (uint32_t)r) && leaves(px, py, holder & 0x7FFu);
                        // ONE entry per event: a near miss records the candidate; a near undercut records the holder it replaces (the
                        // candidate holds the key now, and is recorded in its turn when it loses it: "Exactness") -- if the holder is this
                        // wave's OWN candidate: after the pooling every wave starts from the same holder, and eight waves undercutting it
                        // would record it eight times (first version: lists overflowed, tiles went through the literal kernel).  The wave
                        // that owns a holder either still holds it at the end (recorded then), or has replaced it itself: within the window
                        // (recorded there), or by more than the window -- and then it lies outside the final minimum's bracket.
                        if (key[i] < holder && (holder & 0x7FFu) % (uint32_t)kStripWaves != (uint32_t)wave) skip = true;
                        if (!skip) strip_record(sCnt + lane * kOut + i, sList + (lane * kOut + i) * kStripListCap, max(key[i], holder), min(key[i], holder));
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kOut; ++i) mn[i] = min(mn[i], key[i]);
    }

    // ---- the eight waves' minima pooled per pixel; a wave whose own holder lies within kNearD of the pooled one records it
    if (laneOk) {
#pragma unroll
        for (int i = 0; i < kOut; ++i) atomicMin(&sGmin[lane * kOut + i], mn[i]);
    }
    __syncthreads();
    if (laneOk) {
#pragma unroll
        for (int i = 0; i < kOut; ++i) adopt(i, sGmin[lane * kOut + i]);
    }
    __syncthreads();
    // ---- per pixel: a zero-cost holder IS the answer, and so is a holder with no recorded key within kNearD of it (what was recorded
    // against earlier, larger minima is stale); the others go through the chain; an overflowed list flags the tile
    for (int e = tid; e < kStripSlots; e += kStripThreads) {
        const int l = e / kOut, i = e - l * kOut;
        if (l >= kStripPix || L0 + l >= laneSize) continue;
        const uint32_t g = sGmin[e], n = sCnt[e];
        sFinal[e] = (uint16_t)(g & 0x7FFu);
        if (g < 2048u || n == 0u) continue;
        uint32_t live = 0u;
        for (uint32_t k = 0; k < min(n & 0x7FFFFFFFu, (uint32_t)kStripListCap); ++k) {
            const uint32_t key = sList[e * kStripListCap + k];
            live += (key >= g && key - g <= kNearD && key != g) ? 1u : 0u;
        }
        if ((n >> 31) != 0u) {               // more candidates within reach than the list holds: exact ties (flat or periodic content)
            const int px = kTall ? I0 + i : L0 + l, py = kTall ? L0 + l : I0 + i;
            if (atomicExch(&a.tileFlags[(py / kTH) * a.flagTilesX + px / kTW], 1u) == 0u) {
                const uint32_t slot = atomicAdd(a.flagged, 1u);
                if (slot < (uint32_t)kShareBelow) a.flagged[1 + slot] = (uint32_t)((py / kTH) * a.flagTilesX + px / kTW);
            }
            continue;
        }
        if (live == 0u) continue;
        sHard[atomicAdd(&sHardN, 1u)] = (uint16_t)e;
    }
    __syncthreads();
    // ---- the shader's own chain (motion.comp:33-47) for the candidates of a pixel that came within reach of its minimum: the wave
    // takes one pixel at a time, lane j computes the distance of block position j, every lane adds the 64 in the shader's order
    const uint32_t hardN = sHardN;
    for (uint32_t hI = (uint32_t)wave; hI < hardN; hI += (uint32_t)kStripWaves) {
        const int e = (int)sHard[hI];
        const int l = e / kOut, i = e - l * kOut;
        const int qx = kTall ? I0 + i : L0 + l, qy = kTall ? L0 + l : I0 + i;
        const uint32_t g = sGmin[e];
        const uint32_t n = min(sCnt[e] & 0x7FFFFFFFu, (uint32_t)kStripListCap);
        // lane 0: the pooled holder; lanes 1 .. n: the recorded keys that lie within reach of it
        uint32_t mine = 0xFFFFFFFFu;
        if (lane == 0) mine = g;
        else if ((uint32_t)lane <= n) { const uint32_t k = sList[e * kStripListCap + lane - 1]; mine = (k >= g && k - g <= kNearD) ? k : 0xFFFFFFFFu; }
        const int cx = qx - kB / 2 + (lane & 7), cy = qy - kB / 2 + (lane >> 3);
        const bool posIn = cx >= 0 && cx < W && cy >= 0 && cy < H;
        const uint32_t ctex = posIn ? *reinterpret_cast<const uint32_t *>(a.curr + (size_t)cy * (size_t)a.currPitch + (size_t)cx * 4u) : 0u;
        const float cf[4] = {unorm8_to_float(byte0(ctex)), unorm8_to_float(byte1(ctex)), unorm8_to_float(byte2(ctex)), unorm8_to_float(byte3(ctex))};
        float bestV = __builtin_inff();
        uint32_t bestR = 0xFFFFFFFFu, done = 0xFFFFFFFFu;
        unsigned long long todo = __ballot(mine != 0xFFFFFFFFu);
        while (todo != 0ull) {
            const int b = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t rank = (uint32_t)__builtin_amdgcn_readlane((int)mine, b) & 0x7FFu;
            if (rank == done) continue;                                // (the same candidate recorded twice in a row)
            done = rank;
            const int scan = (int)a.rank2scan[rank];
            const int dy = scan / kSide - kR, dx = scan % kSide - kR;
            const int sx = cx + dx, sy = cy + dy;
            uint32_t ptex = 0u;
            if (posIn && sx >= 0 && sy >= 0 && sx < W && sy < H)
                ptex = *reinterpret_cast<const uint32_t *>(a.prev + (size_t)sy * (size_t)a.prevPitch + (size_t)sx * 4u);
            const f32x4 pp = {unorm8_to_float(byte0(ptex)), unorm8_to_float(byte1(ptex)), unorm8_to_float(byte2(ptex)), unorm8_to_float(byte3(ptex))};
            sDist[wave][lane] = posIn ? dist4<true>(cf, pp) : 0.0f;    // a position outside the image adds +0.0f: the sum unchanged
            wave_lds_sync();
            float v = 0.0f;
#pragma unroll
            for (int k = 0; k < kB * kB; ++k) v += sDist[wave][k];
            wave_lds_sync();
            // a candidate whose block leaves prev stands for its plateau: the first member in tie order takes its place
            uint32_t rr = rank;
            if (block_leaves_prev(qx, qy, dx, dy, W, H)) {
                for (uint32_t r0 = 0; r0 < rank; r0 += 64u) {
                    const uint32_t t = r0 + (uint32_t)lane;
                    const unsigned long long hit = __ballot(t < rank && leaves(qx, qy, t));
                    if (hit != 0ull) { rr = r0 + (uint32_t)__builtin_ctzll(hit); break; }
                }
            }
            if (v < bestV || (v == bestV && rr < bestR)) { bestV = v; bestR = rr; }
        }
        if (lane == 0) sFinal[e] = (uint16_t)bestR;
    }
    __syncthreads();
    // ---- the vectors, and the rows / columns this item has decided
    for (int e = tid; e < kStripSlots; e += kStripThreads) {
        const int l = e / kOut, i = e - l * kOut;
        if (l >= kStripPix || L0 + l >= laneSize) continue;
        const int px = kTall ? I0 + i : L0 + l, py = kTall ? L0 + l : I0 + i;
        const int scan = (int)a.rank2scan[sFinal[e]];
        const int dyi = scan / kSide, dxi = scan - dyi * kSide;
        *reinterpret_cast<uint16_t *>(a.mv + (size_t)py * (size_t)a.mvPitch + (size_t)px * 2u) =
            (uint16_t)(uint8_t)(int8_t)(dxi - kR) | (uint16_t)((uint16_t)(uint8_t)(int8_t)(dyi - kR) << 8);
    }
    if (tid < kStripPix && L0 + tid < laneSize) atomicOr(&(kTall ? a.colBand : a.rowBand)[L0 + tid], bit);
}

// item = band (left, right, top, bottom) x chunk of 57 pixels along it; which bands are worked on is read off the call's top hint
template <bool kRankIsScan>
__global__ __launch_bounds__(kStripThreads) void motion_strip_kernel(StripArgs a) {
    constexpr int kPmax = (kStripCols + kB - 1 + 2 * kR) | 1;
    __shared__ uint32_t sWin[kStripWinLane * kPmax];                   // 19.6 KB
    __shared__ uint32_t sGmin[kStripSlots], sCnt[kStripSlots];        // 3 + 3 KB
    __shared__ uint32_t sList[kStripSlots * kStripListCap];            // 18 KB
    __shared__ uint16_t sFinal[kStripSlots], sHard[kStripSlots];      // 1.5 + 1.5 KB
    __shared__ float sDist[kStripWaves][kB * kB];                      // 4 KB
    __shared__ uint16_t sScan[kRankIsScan ? 2 : kCand + 1];
    __shared__ uint32_t sHardN;
    const int item = (int)blockIdx.x;
    const int side = item < 2 * a.chunksTall ? item / a.chunksTall : 2 + (item - 2 * a.chunksTall) / a.chunksWide;
    const int chunk = item < 2 * a.chunksTall ? item % a.chunksTall : (item - 2 * a.chunksTall) % a.chunksWide;
    // the call's top hint: entry 0 of its order = rank | window offset << 16, offset = (dx + R) * kWinH + (dy + R)
    const uint32_t top = a.order32[0];
    const int dx = (int)((top >> 16) / (uint32_t)kWinH) - kR, dy = (int)((top >> 16) % (uint32_t)kWinH) - kR;
    // Content that moved by (-dx, -dy) exposes the left edge when dx < 0, the right when dx > 0, the bottom when dy > 0, the top when
    // dy < 0: pixels without any match, every candidate summed in full whoever does it.  (The OPPOSITE edges match nearly -- an
    // upscaler filters the rows and columns next to the border differently -- and there the persistent kernel's partial-distortion
    // walks are cheaper than this kernel's full search: measured, round 5.)
    const bool active = side == 0 ? dx < 0 : side == 1 ? dx > 0 : side == 2 ? dy < 0 : dy > 0;
    if (!active) return;
    if (!kRankIsScan) {
        for (int i = threadIdx.x; i < kCand; i += kStripThreads) sScan[i] = (uint16_t)a.rank2scan[i];
        __syncthreads();
    }
    const int L0 = chunk * kStripPix;
#define LFG_STRIP_ITEM(tall, across, edge, i0) strip_item<tall, across, edge, kRankIsScan>(a, L0, i0, 1u << (side & 1), sWin, sGmin, sCnt, sList, sFinal, sHard, sDist, sScan, sHardN)
    if (side == 0) LFG_STRIP_ITEM(true, kStripCols, 0, 0);
    else if (side == 1) LFG_STRIP_ITEM(true, kStripCols, 1, a.W - kStripCols);
    else if (side == 2) LFG_STRIP_ITEM(false, kStripRows, 0, 0);
    else LFG_STRIP_ITEM(false, kStripRows, 1, a.H - kStripRows);
#undef LFG_STRIP_ITEM
}

bool strip_frames_ok(const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv) {
    return curr.width >= 256u && curr.height >= 256u && (mv.pitch & 1u) == 0u && ((uintptr_t)mv.data & 1u) == 0u;
}

hipError_t launch_motion_strip(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv, const uint32_t *order32,
                               const uint32_t *rank2scan, bool rankIsScan, uint32_t *colBand, uint32_t *rowBand,
                               uint32_t *tileFlags, uint32_t *flagged, int flagTilesX, int ldsPad) {
    StripArgs a;
    a.prev = (const uint8_t *)prev.data; a.curr = (const uint8_t *)curr.data;
    a.prevPitch = (int)prev.pitch; a.currPitch = (int)curr.pitch; a.W = (int)curr.width; a.H = (int)curr.height;
    a.mv = (int8_t *)mv.data; a.mvPitch = (int)mv.pitch;
    a.order32 = order32; a.rank2scan = rank2scan; a.colBand = colBand; a.rowBand = rowBand;
    a.tileFlags = tileFlags; a.flagged = flagged; a.flagTilesX = flagTilesX;
    a.chunksTall = (a.H + kStripPix - 1) / kStripPix; a.chunksWide = (a.W + kStripPix - 1) / kStripPix;
    const int items = 2 * a.chunksTall + 2 * a.chunksWide;
    // (ldsPad: dynamic LDS nobody touches -- above 80 KB a CU holds ONE of these workgroups, and the items spread over the chip
    //  instead of sharing CUs two by two while others idle)
    if (rankIsScan) hipLaunchKernelGGL(motion_strip_kernel<true>, dim3((unsigned)items), dim3(kStripThreads), (size_t)ldsPad, s, a);
    else hipLaunchKernelGGL(motion_strip_kernel<false>, dim3((unsigned)items), dim3(kStripThreads), (size_t)ldsPad, s, a);
    return hipGetLastError();
}

}  // namespace lfg
