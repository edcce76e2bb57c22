// motion_order.hip -- the candidate tables of the 8 / 16 motion paths and this call's visiting order (shaders/motion.comp:27-28
// visits the candidates in scan order; the ORDER changes nothing but what the prefilter records along the way).
#include "lfg_motion_common.hpp"

#include <algorithm>

namespace lfg {

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
    // ... and (round 5) that any 32 consecutive entries read 32 different LDS banks.  A lattice test reads the window at the
    // candidate's offset (dx+R)*kWinH + (dy+R) plus constants, one ds_read_b32 per point: its bank is that offset mod 32, conflicts count
    // within each half of the wave, and 32 random offsets put three or four on the busiest bank -- SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS was
    // 2.8 since round 2, "the pseudo-random order's lattice reads".  So the permutation is dealt out by bank: the candidates of each of
    // the 32 classes offset mod 32 keep their pseudo-random order among themselves, and the visiting order takes one from every class in
    // turn -- entry j has class j mod 32 (classes that run out are skipped: the last few entries), so every window of 32 entries holds each
    // class once wherever it starts; the hints a call moves to the front leave a duplicate or two.  Still "unrelated to position": which
    // member of a class comes when is the shuffle's.
    if (LFG_ORDER_BY_BANK) {
        uint16_t byClass[32][kCand / 32 + 2];
        int have[32] = {0};
        for (int i = 1; i < kCand; ++i) {
            const int scan = order[i], cls = ((scan % kSide) * kWinH + scan / kSide) & 31;
            byClass[cls][have[cls]++] = (uint16_t)scan;
        }
        int at = 1;
        for (int k = 0; at < kCand; ++k)
            for (int cls = 0; cls < 32; ++cls)
                if (k < have[cls]) order[at++] = byClass[cls][k];
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
constexpr int kHintThreadsInFlight = LFG_HINT_THREADS;
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
    __shared__ uint32_t sWaveSum[kWaves], sWaveClose[kWaves], sWaveModerate[kWaves];
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
    // (moderate: a best SAD of 310 - 2,200 -- a match under sensor noise of +-3 levels and more at the 1080p input: the true candidate's SAD
    //  is 228 - 292 (5 % - 95 % of 1,024 blocks) at +-2, 322 - 413 at +-3, 416 - 532 at +-4, 778 - 997 at +-8, 1,161 - 1,477 at +-12
    //  (tools/hint_sad_by_noise.py) -- where the persistent kernel's variant with the walks by SADs pays, motion_prefilter_kernel<false, 1>:
    //  +15 % at +-3, +27 % at +-4 .. +-12, +5 % at +-16, and -3 % at +-2 and below, -4 % at +-24: hence 310 and 2,200)
    const uint32_t moderate = (uint32_t)__popcll(__ballot((hint >> 11) >= 310u && (hint >> 11) < 2200u));
    if (lane == 0) { sWaveSum[wv] = unmatched; sWaveClose[wv] = close; sWaveModerate[wv] = moderate; }
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
        uint32_t moderateAll = 0u;
        for (int w = 0; w < kWaves; ++w) moderateAll += sWaveModerate[w];
        // (bit 0: the verdict; above it the two counts, for LFG_DEBUG; bit 29: half the sample blocks or more match moderately well -- the
        //  variant of the persistent kernel for the lane's next call; bit 30 is the literal pass's; bit 31: most sample blocks have a match
        //  -- the host sizes the persistent grid of the lane's next call by it)
        order32[kCand + 2] = (((closeAll & 0xFFFFu) * 16u >= 15u * (uint32_t)kHints) ? 1u : 0u) | ((closeAll & 0xFFFFu) << 1) | (((closeAll >> 16) & 0x7FFu) << 12) |
                             ((moderateAll * 2u >= (uint32_t)kHints ? 1u : 0u) << 29) | (mostMatch << 31);
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

// Both launches of a call's visiting order: hints (which also clears the call's control area, `clearWords` words from
// `clearFrom`), then the order itself into callOrder[0 .. kCand + 2].  hints: kHints words of scratch.
hipError_t launch_motion_order(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, uint32_t *hints, uint32_t *callOrder,
                               const uint32_t *entryOfScan, const uint32_t *baseScan, uint32_t *clearFrom, int clearWords, bool framesInFlight) {
    // (256 threads a workgroup beside the prefilter launches of other frames in flight -- see the kernel --, 1,024 alone)
    if (framesInFlight)
        hipLaunchKernelGGL(motion_hint_kernel<kHintThreadsInFlight>, dim3(kHints), dim3(kHintThreadsInFlight), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int)curr.width, (int)curr.height, hints, clearFrom, clearWords);
    else
        hipLaunchKernelGGL(motion_hint_kernel<kHintThreadsAlone>, dim3(kHints), dim3(kHintThreadsAlone), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (int)curr.width, (int)curr.height, hints, clearFrom, clearWords);
    hipLaunchKernelGGL(motion_order_kernel, dim3(1), dim3(kHints), 0, s, hints, baseScan, entryOfScan, callOrder);
    return hipGetLastError();
}

}  // namespace lfg
