// motion_strip.hip -- the strips a pan exposes (shaders/motion.comp:27-52 for pixels that have NO match).
//
// A frame that moves as a whole exposes new content along one or two edges: sixteen pixel columns on the left of the 4K
// benchmark frame, eight rows at its bottom.  No candidate matches there, no partial-distortion test can drop anything, and
// every one of the 1,089 candidates has to be summed in full for every such pixel.  The persistent kernel did that inside its
// 1,800-line work unit (narrow search, row band: prefilter_narrow.inc, prefilter_rowband.inc) at 256 VGPRs and two waves per
// SIMD: 262 units of 225 - 232 us, two thirds of a frame's motion work.  This kernel is that search
// written for what it is -- EVERY candidate, a few pixels -- and nothing else:
//
//   * one work item = 57 pixels ALONG an edge x the band ACROSS it (16 columns, or 8 rows) x all candidates; a workgroup of
//     eight waves, wave w taking the ranks = w mod 8 in ascending order;
//   * lane = block position along the edge (64 positions serve 57 pixels), so all 64 lanes compute distances for every
//     candidate: kIn = band + 7 distances along the in-lane axis, an in-lane pairwise tree to the band's 8-sums, and the
//     8-sum ACROSS lanes by DPP wavefront shifts (v_add_f32_dpp wave_shl:1: lane l adds lane l + 1) -- no transposition, no
//     slab, no LDS traffic but the window reads; S~ is again a depth-6 pairwise tree over exactly rounded integer squared
//     distances and v_sqrt_f32: within 9 u of 255 x the real-arithmetic cost (motion_prefilter.hip, "Bracket");
//   * per pixel ONE register: key = (12 mantissa bits of S~ | rank), kept as the running minimum by v_min_u32; a second
//     instruction per pixel (v_sad_u32) measures how close the candidate came to the holder, and only when some pixel of the
//     wave saw a near miss or a near undercut (|key - holder| <= kNearD: a superset of "within the bracket") does a scalar
//     branch record both in the pixel's short list.  Every candidate within the bracket of the final minimum is then either a
//     wave's final holder or recorded (proof below), and the shader's own 64-term chain decides among those few;
//   * ~130 VGPRs, 62 KB of LDS: two workgroups (four waves per SIMD) per CU.
//
// The kernel decides its pixels COMPLETELY -- vectors written, no lists, no resolve kernel -- and says which pixels it took in
// two small tables (colBand[y]: columns lo..hi of row y; rowBand[x]: rows lo..hi of column x) that the persistent kernel's rim
// units and the resolve kernel read: those pixels are parked there like pixels outside the image.  Which strips exist is read
// off the call's top hint (a vector with dx < 0 exposes the left edge, dy > 0 the bottom ...): a wrong guess costs time, never
// a result -- pixels the persistent kernel does not find in the tables it searches itself, as before.
//
// Exactness.  K(m) = (T(m) << 11-ish | rank), T = S~ truncated to 12 mantissa bits, so T <= S~ < T (1 + 2^-12).  Let C be the
// candidates with S~(m) <= kRatio min S~ -- the only ones that can be the shader's first strict minimum ("Bracket").  For m in C
// and b the holder of the smallest key: T(m) <= kRatio min S~ < kRatio (1 + 2^-12) T(b), i.e. less than 2.7 units of T's last
// place above T(b) whatever the binade: K(m) - K(b) <= 2 * 2048 + 2047 < kNearD.  A wave visits its candidates in ascending rank;
// x is NEAR when |K(x) - K(holder)| <= kNearD; a near MISS records x, a near UNDERCUT records the holder that x replaces.  Take m in C, m != b, in b's wave:
// visited after b, it is near b; visited before, it is either near the holder of that moment (recorded), or more than kNearD
// above it (then above K(b) + kNearD: not in C), or becomes the holder unrecorded -- but then every later change of holder on
// the way down to b happens within kNearD (all those keys lie between K(b) and K(m)) and records the holder it replaces, m
// first; and a near undercut by m itself makes m the holder, recorded likewise when it loses the key.  Candidates of C in another wave: the same argument with that wave's own minimum, whose holder is compared with the
// pooled minimum at the end.  S~ = 0 means cost exactly 0: the key IS the rank, the smallest rank wins, no chain.  Candidates
// whose block leaves prev altogether tie exactly (a plateau, motion_prefilter.hip): same T, the smallest rank holds the key,
// and the chain phase maps every candidate to the first member of its plateau as the resolve kernel does.
#include "lfg_motion_common.hpp"

#include <algorithm>
#include <vector>

namespace lfg {

constexpr int kStripPix = 64 - (kB - 1);            // 57 pixels along the lane axis per item
constexpr int kStripThreads = 512, kStripWaves = kStripThreads / 64;
constexpr int kStripWinLane = 64 + 2 * kR;          // 96 window positions along the lane axis
constexpr int kStripCols = 16, kStripRows = 8;      // the band across the edge: pixel columns of a left / right item, rows of a top / bottom one
constexpr int kStripListCap = 6;                    // recorded keys per pixel; more: the tile goes through the literal kernel
constexpr uint32_t kKeyMask = 0xFFFFF800u;          // 12 mantissa bits of S~ above an 11-bit rank
constexpr uint32_t kNearD = 3u * 2048u + 2047u;     // see "Exactness"
constexpr int kStripSlots = 64 * kStripCols;        // pixel slots of an item in LDS (lane * band + i)

typedef const __attribute__((address_space(3))) uint32_t *strip_lds_u32;

// lane l <- lane l + 1 (the last lane reads 0): v_mov_b32_dpp / folded into the add that uses it
__device__ __forceinline__ float strip_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
// |a - b| of two unsigned words in one instruction (hipcc has builtins for the byte and halfword SADs only)
__device__ __forceinline__ uint32_t strip_absdiff(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
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
    uint32_t *colBand, *rowBand;                 // [H], [W]: lo | hi << 16 | 1 << 31 -- cleared with the call's control area
    uint32_t *tileFlags, *flagged; int flagTilesX;   // the literal kernel's tile flags, count and list (motion_prefilter.hip: gaveUp)
    int chunksTall, chunksWide;                  // items per side
};

// One item.  kTall: the band is kOut pixel COLUMNS (lane axis = y, in-lane axis = x); else kOut pixel ROWS (lane axis = x).
template <bool kTall, int kOut, bool kRankIsScan>
__device__ __forceinline__ void strip_item(const StripArgs &a, const int L0, const int I0, uint32_t *sWin, uint32_t *sGmin, uint32_t *sCnt, uint32_t *sList,
                                           uint16_t *sFinal, uint16_t *sHard, float (*sDist)[kB * kB], const uint16_t *sScan, uint32_t &sHardN) {
    constexpr int kIn = kOut + kB - 1;               // block positions per lane
    constexpr int kWinIn = kIn + 2 * kR;             // window extent along the in-lane axis
    constexpr int kP = kWinIn | 1;                   // window pitch (words): odd, so the lanes of a read fall into distinct banks
    static_assert(kStripWinLane * kP <= kStripWinLane * ((kStripCols + kB - 1 + 2 * kR) | 1), "window fits");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = a.W, H = a.H;
    const int laneSize = kTall ? H : W;              // image extent along the lane axis
    // image coordinates of (lane-axis position u, in-lane position v) of the WINDOW: L0 - 4 - R + u, I0 - 4 - R + v
    const int wl0 = L0 - kB / 2 - kR, wi0 = I0 - kB / 2 - kR;

    // ---- this lane's kIn block positions of the current frame (lane-axis coordinate L0 - 4 + lane), zero and invalid outside the image
    uint32_t c[kIn], cc[kIn];
    uint32_t valid = 0u;
    {
        const int gl = L0 - kB / 2 + lane;
        const bool okl = gl >= 0 && gl < laneSize;
#pragma unroll
        for (int j = 0; j < kIn; ++j) {
            const int gi = I0 - kB / 2 + j;
            const int x = kTall ? gi : gl, y = kTall ? gl : gi;
            const bool ok = okl && gi >= 0 && gi < (kTall ? W : H);
            const uint32_t t = *reinterpret_cast<const uint32_t *>(a.curr + (size_t)clampi(y, 0, H - 1) * (size_t)a.currPitch + (size_t)clampi(x, 0, W - 1) * 4u);
            c[j] = ok ? t : 0u;
            valid |= (ok ? 1u : 0u) << j;
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
    for (int j = 0; j < kIn; ++j) cc[j] = __builtin_amdgcn_udot4(c[j], c[j], 0x4B000000u, false);     // 2^23 + |c|^2 as float bits
    __syncthreads();

    // this lane's pixels: lane-axis coordinate L0 + lane (lanes 0 .. 56), in-lane coordinates I0 .. I0 + kOut - 1
    const bool laneOk = lane < kStripPix && L0 + lane < laneSize;
    const int allValid = __builtin_amdgcn_readfirstlane((int)(__ballot(valid != (1u << kIn) - 1u) == 0ull));
    uint32_t mn[kOut];
#pragma unroll
    for (int i = 0; i < kOut; ++i) mn[i] = 0xFFFFFFFFu;
    const strip_lds_u32 winLane = (strip_lds_u32)sWin + lane * kP;
    auto leaves = [&](int px, int py, uint32_t rank) {
        const uint32_t sc = kRankIsScan ? rank : (uint32_t)sScan[min(rank, (uint32_t)kCand - 1u)];
        const int dyi = (int)((sc * 1986u) >> 16), dxi = (int)sc - 33 * dyi;
        return block_leaves_prev(px, py, dxi - kR, dyi - kR, W, H);
    };

    // ---- every candidate of this wave's share, ascending in rank
    for (int r = wave; r < kCand; r += kStripWaves) {
        const uint32_t sc = kRankIsScan ? (uint32_t)r : (uint32_t)sScan[r];
        const uint32_t dyi = (sc * 1986u) >> 16, dxi = sc - 33u * dyi;        // sc / 33, sc % 33 for sc < 1089
        const uint32_t off = (kTall ? dyi : dxi) * (uint32_t)kP + (kTall ? dxi : dyi);
        const strip_lds_u32 w = winLane + off;
        uint32_t p[kIn];
#pragma unroll
        for (int j = 0; j < kIn; ++j) p[j] = w[j];
        // distances: n = |c|^2 + |p|^2 - 2 c.p by dot products onto float bit patterns (motion_prefilter.hip: columnSums)
        float d[kIn];
#pragma unroll
        for (int j = 0; j < kIn; ++j) {
            const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(p[j], p[j], cc[j], false));
            const float f2 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(c[j], p[j], 0x4B800000u, false));
            d[j] = __builtin_amdgcn_sqrtf((f1 - f2) + 8388608.0f);
        }
        if (!allValid) {                   // (wave-uniform) a position outside the image is skipped by the shader: adds 0 here
            asm volatile("; positions outside the image (strip)");
#pragma unroll
            for (int j = 0; j < kIn; ++j) d[j] = ((valid >> j) & 1u) ? d[j] : 0.0f;
        }
        float a2[kIn - 1], a4[kIn - 3], h8[kOut];
#pragma unroll
        for (int j = 0; j < kIn - 1; ++j) a2[j] = d[j] + d[j + 1];
#pragma unroll
        for (int j = 0; j < kIn - 3; ++j) a4[j] = a2[j] + a2[j + 2];
#pragma unroll
        for (int i = 0; i < kOut; ++i) h8[i] = a4[i] + a4[i + 4];
        // S~ of pixel (lane, i): the 8-sum across lanes, level by level over all kOut values (a DPP instruction that reads a register
        // written just before it waits two states: one chain after the other was a third s_nop)
        float t2[kOut], t4[kOut], u1[kOut], u2[kOut], u3[kOut];
#pragma unroll
        for (int i = 0; i < kOut; ++i) t2[i] = h8[i] + strip_shl1(h8[i]);
#pragma unroll
        for (int i = 0; i < kOut; ++i) u1[i] = strip_shl1(t2[i]);
#pragma unroll
        for (int i = 0; i < kOut; ++i) t4[i] = t2[i] + strip_shl1(u1[i]);
#pragma unroll
        for (int i = 0; i < kOut; ++i) u1[i] = strip_shl1(t4[i]);
#pragma unroll
        for (int i = 0; i < kOut; ++i) u2[i] = strip_shl1(u1[i]);
#pragma unroll
        for (int i = 0; i < kOut; ++i) u3[i] = strip_shl1(u2[i]);
        // ... its key, and how near the key came to the holder's
        uint32_t key[kOut], nearMin = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < kOut; ++i) {
            const float s = t4[i] + strip_shl1(u3[i]);
            key[i] = (__builtin_bit_cast(uint32_t, s) & kKeyMask) | (uint32_t)r;
            nearMin = min(nearMin, strip_absdiff(key[i], mn[i]));
        }
        nearMin = laneOk ? nearMin : 0xFFFFFFFFu;
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(nearMin <= kNearD) != 0ull))) {
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
                        bool skip = key[i] < 2048u && holder < 2048u;
                        if (!skip && (key[i] >> 11) == (holder >> 11)) skip = leaves(px, py, (uint32_t)r) && leaves(px, py, holder & 0x7FFu);
                        if (!skip) {
                            // ONE entry per event: a near miss records the candidate; a near undercut records the holder it replaces
                            // (the candidate holds the key now, and is recorded in its turn when it loses it: "Exactness")
                            const uint32_t at = atomicAdd(&sCnt[lane * kOut + i], 1u);
                            if (at < (uint32_t)kStripListCap) sList[(lane * kOut + i) * kStripListCap + at] = key[i] > holder ? key[i] : holder;
                        }
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
        for (int i = 0; i < kOut; ++i) {
            const uint32_t g = sGmin[lane * kOut + i];
            if (mn[i] != g && mn[i] - g <= kNearD && !(mn[i] < 2048u && g < 2048u)) {
                const int px = kTall ? I0 + i : L0 + lane, py = kTall ? L0 + lane : I0 + i;
                const bool plateau = (mn[i] >> 11) == (g >> 11) && leaves(px, py, mn[i] & 0x7FFu) && leaves(px, py, g & 0x7FFu);
                if (!plateau) {
                    const uint32_t at = atomicAdd(&sCnt[lane * kOut + i], 1u);
                    if (at < (uint32_t)kStripListCap) sList[(lane * kOut + i) * kStripListCap + at] = mn[i];
                }
            }
        }
    }
    __syncthreads();
    // ---- per pixel: a zero-cost holder or an empty list IS the answer; a short list goes through the chain; a full one flags the tile
    for (int e = tid; e < kStripSlots; e += kStripThreads) {
        const int l = e / kOut, i = e - l * kOut;
        if (l >= kStripPix || L0 + l >= laneSize || i >= kOut) continue;
        const uint32_t g = sGmin[e], n = sCnt[e];
        sFinal[e] = (uint16_t)(g & 0x7FFu);
        if (g < 2048u || n == 0u) continue;
        if (n > (uint32_t)kStripListCap) {           // more candidates within reach than the list holds: exact ties (flat or periodic content)
            const int px = kTall ? I0 + i : L0 + l, py = kTall ? L0 + l : I0 + i;
            if (atomicExch(&a.tileFlags[(py / kTH) * a.flagTilesX + px / kTW], 1u) == 0u) {
                const uint32_t slot = atomicAdd(a.flagged, 1u);
                if (slot < (uint32_t)kShareBelow) a.flagged[1 + slot] = (uint32_t)((py / kTH) * a.flagTilesX + px / kTW);
            }
            continue;
        }
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
        const uint32_t n = min(sCnt[e], (uint32_t)kStripListCap);
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
    if (tid < kStripPix && L0 + tid < laneSize)
        (kTall ? a.colBand : a.rowBand)[L0 + tid] = (uint32_t)I0 | ((uint32_t)(I0 + kOut - 1) << 16) | 0x80000000u;
}

// item = side (left, right, top, bottom) x chunk of 57 pixels along it; which sides are worked on is read off the call's top hint
template <bool kRankIsScan>
__global__ __launch_bounds__(kStripThreads, 4) void motion_strip_kernel(StripArgs a) {
    constexpr int kPmax = (kStripCols + kB - 1 + 2 * kR) | 1;
    __shared__ uint32_t sWin[kStripWinLane * kPmax];                   // 21 KB
    __shared__ uint32_t sGmin[kStripSlots], sCnt[kStripSlots];        // 4 + 4 KB
    __shared__ uint32_t sList[kStripSlots * kStripListCap];            // 24 KB
    __shared__ uint16_t sFinal[kStripSlots], sHard[kStripSlots];      // 2 + 2 KB
    __shared__ float sDist[kStripWaves][kB * kB];                      // 2 KB
    __shared__ uint16_t sScan[kRankIsScan ? 2 : kCand + 1];
    __shared__ uint32_t sHardN;
    const int item = (int)blockIdx.x;
    const int side = item < 2 * a.chunksTall ? item / a.chunksTall : 2 + (item - 2 * a.chunksTall) / a.chunksWide;
    const int chunk = item < 2 * a.chunksTall ? item % a.chunksTall : (item - 2 * a.chunksTall) % a.chunksWide;
    // the call's top hint: entry 0 of its order = rank | window offset << 16, offset = (dx + R) * kWinH + (dy + R)
    const uint32_t top = a.order32[0];
    const int dx = (int)((top >> 16) / (uint32_t)kWinH) - kR, dy = (int)((top >> 16) % (uint32_t)kWinH) - kR;
    // content that moved by (-dx, -dy) exposes the left edge when dx < 0, the right when dx > 0, the bottom when dy > 0, the top when dy < 0
    const bool active = side == 0 ? dx < 0 : side == 1 ? dx > 0 : side == 2 ? dy < 0 : dy > 0;
    if (!active) return;
    if (!kRankIsScan) {
        for (int i = threadIdx.x; i < kCand; i += kStripThreads) sScan[i] = (uint16_t)a.rank2scan[i];
        __syncthreads();
    }
    const int L0 = chunk * kStripPix;
    if (side < 2) strip_item<true, kStripCols, kRankIsScan>(a, L0, side == 0 ? 0 : a.W - kStripCols, sWin, sGmin, sCnt, sList, sFinal, sHard, sDist, sScan, sHardN);
    else strip_item<false, kStripRows, kRankIsScan>(a, L0, side == 2 ? 0 : a.H - kStripRows, sWin, sGmin, sCnt, sList, sFinal, sHard, sDist, sScan, sHardN);
}

bool strip_frames_ok(const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv) {
    return curr.width >= 256u && curr.height >= 256u && (mv.pitch & 1u) == 0u && ((uintptr_t)mv.data & 1u) == 0u;
}

hipError_t launch_motion_strip(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv, const uint32_t *order32,
                               const uint32_t *rank2scan, bool rankIsScan, uint32_t *colBand, uint32_t *rowBand,
                               uint32_t *tileFlags, uint32_t *flagged, int flagTilesX) {
    StripArgs a;
    a.prev = (const uint8_t *)prev.data; a.curr = (const uint8_t *)curr.data;
    a.prevPitch = (int)prev.pitch; a.currPitch = (int)curr.pitch; a.W = (int)curr.width; a.H = (int)curr.height;
    a.mv = (int8_t *)mv.data; a.mvPitch = (int)mv.pitch;
    a.order32 = order32; a.rank2scan = rank2scan; a.colBand = colBand; a.rowBand = rowBand;
    a.tileFlags = tileFlags; a.flagged = flagged; a.flagTilesX = flagTilesX;
    a.chunksTall = (a.H + kStripPix - 1) / kStripPix; a.chunksWide = (a.W + kStripPix - 1) / kStripPix;
    const int items = 2 * a.chunksTall + 2 * a.chunksWide;
    if (rankIsScan) hipLaunchKernelGGL(motion_strip_kernel<true>, dim3((unsigned)items), dim3(kStripThreads), 0, s, a);
    else hipLaunchKernelGGL(motion_strip_kernel<false>, dim3((unsigned)items), dim3(kStripThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace lfg
