// Throughput probe for a candidate-per-lane full evaluation of the motion prefilter's cost bracket (gfx950).
//
// The prefilter's full evaluation (csrc/prefilter_sums.inc: columnSums / transpose / rowSumsAndTest) gives a wave ONE candidate:
// lane = position column, the 8 x 8 sums go through a slab in LDS and come back transposed.  Here a wave takes 64 candidates,
// one per lane: the lane walks the segment's 23 x 63 block positions column by column -- 23 distances, the shared pairwise
// tree down the column, then the horizontal tree kept as a ring of seven partial sums per pixel row -- and compares the
// 16 sums of a pixel column with that column's thresholds, which are the same for every lane (LDS broadcast reads).
// No slab, no transposition, no exposed round trip: the chains of a lane's columns are independent.
// This file measures what that costs per candidate, with thresholds that never pass (mode 0) and with thresholds that
// start at +inf and follow the minima through LDS atomics (mode 1: every pixel without a match), and checks a handful of
// sums against the same tree on the host.
//
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o bench_batch_eval bench_batch_eval.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) uint32_t *lds_ro_u32_ptr;

constexpr int kB = 8, kR = 16, kSide = 2 * kR + 1, kCand = kSide * kSide;
constexpr int kPTW = 56, kSeg = 16, kSegD = kSeg + kB - 1;     // 23
constexpr int kCols = kPTW + kB - 1;                            // 63 position columns
constexpr int kWinW = 95, kWinH = 103;
#ifndef CCLDS
#define CCLDS 1
#endif
#ifndef PREFETCH
#define PREFETCH 1
#endif
#ifndef CSMEM
#define CSMEM 0                      // the current-frame texels by scalar loads from a row-major image (four columns per load), |c|^2 by a dot product
#endif
constexpr int kCP = CCLDS ? 48 : 25;                            // pitch of the current-frame columns in LDS: 24 texels, then the 2^23 + |c|^2 words
constexpr int kListK = 10;
constexpr float kRatio = 1.00008f, kRestart = 0.9997f;

#ifndef UNROLL_COLS
#define UNROLL_COLS 4
#endif

struct Params {
    const uint32_t *win;      // [kWinW][kWinH] window texels (column-major), one window for every workgroup
    const uint32_t *cur;      // [kCols][kCP] current-frame texels
    const uint32_t *__restrict__ curRows;   // [kSegD][64] the same, row-major (CSMEM)
    float *sums;              // mode 2: [kCand][16][56] sums for the check
    uint32_t *lists;          // mode 1: records, [workgroup][wave][16][kListK][56]
    uint32_t *stats;          // [0] passes, [1] records
    int batches, mode;
};

__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// the smallest of a wave's 64 unsigned values, wave-uniform (DPP: see csrc/lfg_device.hpp, wave_max_u32)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    auto step = [](uint32_t x, auto ctrl, auto rowMask) {
        const uint32_t moved = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, decltype(ctrl)::value, decltype(rowMask)::value, 0xF, false);
        return x < moved ? x : moved;
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ __launch_bounds__(256, 2) void batch_eval_kernel(Params P) {
    __shared__ uint32_t sWin[kWinW * kWinH];
    __shared__ __attribute__((aligned(16))) uint32_t sCur[kCols * kCP];
    __shared__ __attribute__((aligned(16))) float sThr[kPTW * 16];            // [pixel column][row pair r][r, r + 8]
    __shared__ uint8_t sCnt[4][kPTW * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kWinW * kWinH; i += 256) sWin[i] = P.win[i];
    for (int i = tid; i < kCols * kCP; i += 256) {
        uint32_t v = P.cur[i];
        if (CCLDS && i % kCP >= 24) { const uint32_t c = P.cur[i - 24]; v = __builtin_amdgcn_udot4(c, c, 0x4B000000u, false); }
        sCur[i] = v;
    }
    for (int i = tid; i < kPTW * 16; i += 256) {
        sThr[i] = P.mode == 1 ? __builtin_inff() : -1.0f;
        for (int w = 0; w < 4; ++w) sCnt[w][i] = 0u;
    }
    __syncthreads();
    uint32_t passes = 0u, records = 0u;
    float check = 0.0f;
    uint32_t *const myList = P.lists + ((size_t)blockIdx.x * 4 + wave) * (size_t)(16 * kListK * kPTW);
    for (int b = 0; b < P.batches; ++b) {
        // this lane's candidate: consecutive ranks (= scan order: dy outer, dx inner), a different stretch per wave
        const int rank = ((b * 4 + wave) * 64 + lane) % kCand;
        const int dy = rank / kSide - kR, dx = rank % kSide - kR;
        const lds_ro_u32_ptr w0 = (lds_ro_u32_ptr)sWin + (dx + kR) * kWinH + (dy + kR);
        const lds_ro_u32_ptr c0 = (lds_ro_u32_ptr)sCur;
        f32x2 V8p[8], H2a[8], H2b[8], H4[4][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            V8p[r] = f32x2{0.0f, 0.0f}; H2a[r] = V8p[r]; H2b[r] = V8p[r];
#pragma unroll
            for (int k = 0; k < 4; ++k) H4[k][r] = V8p[r];
        }
        auto fetch = [&](uint32_t (&p)[kSegD], int x) {
            const lds_ro_u32_ptr w = w0 + min(x, kCols - 1) * kWinH;
#pragma unroll
            for (int j = 0; j < kSegD; ++j) p[j] = w[j];
        };
        auto column = [&](const int x, const uint32_t (&p)[kSegD], const uint2 (&c2)[kSegD], auto slotC) {
            constexpr int slot = decltype(slotC)::value;
            uint32_t c[kSegD], cc[kSegD];
            const lds_ro_u32_ptr cw = c0 + x * kCP;
            if (CSMEM) {
#pragma unroll
                for (int j = 0; j < kSegD; ++j) {
                    c[j] = (slot & 1) ? c2[j].x : c2[j].y;
                    cc[j] = __builtin_amdgcn_udot4(c[j], c[j], 0x4B000000u, false);
                }
            } else {
#pragma unroll
            for (int j = 0; j < kSegD; ++j) { c[j] = cw[j]; cc[j] = CCLDS ? cw[24 + j] : __builtin_amdgcn_udot4(c[j], c[j], 0x4B000000u, false); }
            }
            auto dist2 = [&](int j) {
                const float f1 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(p[j], p[j], cc[j], false));
                const float f2 = __builtin_bit_cast(float, __builtin_amdgcn_udot4(c[j], p[j], 0x4B800000u, false));
                return f32x2{f1, f2};
            };
            constexpr int kPairs = kSegD - 8;                          // 15
            const f32x2 kBias = {8388608.0f, 8388608.0f};
            f32x2 A[kPairs];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 a = dist2(j), bb = dist2(j + 8);
                const f32x2 N = (f32x2{a.x, bb.x} - f32x2{a.y, bb.y}) + kBias;
                A[j] = f32x2{__builtin_amdgcn_sqrtf(N.x), __builtin_amdgcn_sqrtf(N.y)};
            }
            float dHi[kSegD - 16];
#pragma unroll
            for (int k = 0; k + 1 < kSegD - 16; k += 2) {
                const f32x2 a = dist2(16 + k), bb = dist2(17 + k);
                const f32x2 N = (f32x2{a.x, bb.x} - f32x2{a.y, bb.y}) + kBias;
                dHi[k] = __builtin_amdgcn_sqrtf(N.x); dHi[k + 1] = __builtin_amdgcn_sqrtf(N.y);
            }
            { const f32x2 a = dist2(kSegD - 1); dHi[kSegD - 17] = __builtin_amdgcn_sqrtf((a.x - a.y) + 8388608.0f); }
#pragma unroll
            for (int j = 8; j < kPairs; ++j) A[j] = f32x2{A[j - 8].y, dHi[j - 8]};
            f32x2 Bt[kPairs - 1], G[kPairs - 3], C8[8];
#pragma unroll
            for (int j = 0; j < kPairs - 1; ++j) Bt[j] = A[j] + A[j + 1];
#pragma unroll
            for (int j = 0; j < kPairs - 3; ++j) G[j] = Bt[j] + Bt[j + 2];
#pragma unroll
            for (int j = 0; j < 8; ++j) C8[j] = G[j] + G[j + 4];
            // the horizontal tree: h2[x-1] = V8[x-1] + V8[x]; h4[x-3] = h2[x-3] + h2[x-1]; S[x-7] = h4[x-7] + h4[x-3]
            f32x2 S[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const f32x2 h2 = V8p[r] + C8[r];
                V8p[r] = C8[r];
                const f32x2 h4 = H2a[r] + h2;
                H2a[r] = H2b[r]; H2b[r] = h2;
                S[r] = H4[slot][r] + h4;
                H4[slot][r] = h4;
            }
            const int px = max(x - (kB - 1), 0);                       // (the first seven columns complete no pixel column: their sums are garbage, see below)
            if (P.mode == 2) {
                if (x < kB - 1) return;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    P.sums[((size_t)rank * 16 + r) * kPTW + px] = S[r].x;
                    P.sums[((size_t)rank * 16 + r + 8) * kPTW + px] = S[r].y;
                }
                return;
            }
            const f32x2 *const thrP = reinterpret_cast<const f32x2 *>(sThr + px * 16);
            f32x2 T[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) T[r] = thrP[r];
            f32x2 dm = T[0] - S[0];
            float top = __builtin_fmaxf(dm.x, dm.y);
#pragma unroll
            for (int r = 1; r < 8; ++r) { dm = T[r] - S[r]; top = __builtin_fmaxf(top, __builtin_fmaxf(dm.x, dm.y)); }
            check += top;
            if (__builtin_amdgcn_readfirstlane((int)(__ballot(top >= 0.0f) == 0ull || x < kB - 1))) return;
            // some candidate of the batch passes for some pixel of this column
            asm volatile("; a column takes records");
#pragma unroll 1
            for (int rr = 0; rr < 16; ++rr) {
                const int r = rr >> 1, hb = rr & 1;
                float s = 0.0f, t = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) { if (k == r) { s = hb ? S[k].y : S[k].x; t = hb ? T[k].y : T[k].x; } }
                const bool pass = s <= t;
                if (__ballot(pass) == 0ull) continue;
                passes += 1u;
                const float tU = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, t)));
                const uint32_t smin = wave_min_u32(pass ? __builtin_bit_cast(uint32_t, s) : 0x7F800000u);
                const float sminF = __builtin_bit_cast(float, smin);
                uint32_t capBits = __builtin_bit_cast(uint32_t, sminF * kRatio);
                if (smin == 0u) {          // a zero-cost candidate: the first in tie order among those of this batch
                    const int first = __builtin_ctzll(__ballot(pass && s == 0.0f));
                    capBits = 0x00800000u + (uint32_t)__builtin_amdgcn_readlane(rank, first);
                }
                const float capF = __builtin_bit_cast(float, capBits);
                const bool keep = pass && s != 0.0f && s <= capF;
                const unsigned long long km = __ballot(keep);
                uint8_t *const cntW = &sCnt[wave][px * 16 + 2 * r + hb];
                const uint32_t n0 = sminF < tU * kRestart ? 0u : (uint32_t)*(volatile uint8_t *)cntW;       // (every earlier record of this list is dead)
                const uint32_t slot = n0 + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                if (keep) {
                    myList[((2 * r + hb) * kListK + min(slot, (uint32_t)kListK - 1u)) * kPTW + px] = ((__builtin_bit_cast(uint32_t, s) >> 10) << 11) | (uint32_t)rank;
                    records += 1u;
                }
                if (lane == 0) {
                    *(volatile uint8_t *)cntW = (uint8_t)min(n0 + (uint32_t)__builtin_popcountll(km), 255u);
                    __hip_atomic_fetch_min(reinterpret_cast<uint32_t *>(sThr) + px * 16 + 2 * r + hb, capBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        };
        uint32_t pA[kSegD], pB[kSegD];
        fetch(pA, 0);
#pragma unroll 1
        for (int x0 = 0; x0 < kCols + 1; x0 += 4) {                    // 64 columns: the last one is padding (it re-reads column 62 and its sums complete no pixel)
            uint2 c2[kSegD];
            auto currentTexels = [&](int x) {
#if defined(__HIP_DEVICE_COMPILE__)
                if (CSMEM) {
#pragma unroll
                    for (int j = 0; j < kSegD; ++j) c2[j] = *(const __attribute__((address_space(4))) uint2 *)(uintptr_t)(P.curRows + j * 64 + x);     // (constant address space: scalar loads)
                }
#endif
            };
            currentTexels(x0);
            if (PREFETCH) fetch(pB, x0 + 1);
            column(x0, pA, c2, std::integral_constant<int, 1>{});
            if (!PREFETCH) fetch(pB, x0 + 1);
            if (PREFETCH) fetch(pA, x0 + 2);
            column(x0 + 1, pB, c2, std::integral_constant<int, 2>{});
            if (!PREFETCH) fetch(pA, x0 + 2);
            currentTexels(x0 + 2);
            if (PREFETCH) fetch(pB, x0 + 3);
            column(x0 + 2, pA, c2, std::integral_constant<int, 3>{});
            if (!PREFETCH) fetch(pB, x0 + 3);
            if (PREFETCH) fetch(pA, x0 + 4);
            if (x0 + 3 < kCols) column(x0 + 3, pB, c2, std::integral_constant<int, 0>{});
            if (!PREFETCH) fetch(pA, x0 + 4);
        }
    }
    if (check == 12345.678f) P.stats[2] = 1u;
    passes = __builtin_amdgcn_readfirstlane(passes);
    if (lane == 0) atomicAdd(&P.stats[0], passes);
    atomicAdd(&P.stats[1], records);
}

int main(int argc, char **argv) {
    const int batches = argc > 1 ? atoi(argv[1]) : 8;
    const int groups = argc > 2 ? atoi(argv[2]) : 512;
    std::vector<uint32_t> win(kWinW * kWinH), cur(kCols * kCP);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    for (auto &v : win) v = rnd();
    for (auto &v : cur) v = rnd();
    std::vector<uint32_t> curRows(kSegD * 64, 0u);
    for (int col = 0; col < kCols; ++col) for (int row = 0; row < kSegD; ++row) curRows[row * 64 + col] = cur[col * kCP + row];
    uint32_t *dWin, *dCur, *dLists, *dStats, *dCurRows;
    CK(hipMalloc(&dCurRows, curRows.size() * 4));
    CK(hipMemcpy(dCurRows, curRows.data(), curRows.size() * 4, hipMemcpyHostToDevice));
    float *dSums;
    CK(hipMalloc(&dWin, win.size() * 4)); CK(hipMalloc(&dCur, cur.size() * 4));
    CK(hipMalloc(&dSums, (size_t)kCand * 16 * kPTW * 4));
    CK(hipMalloc(&dLists, (size_t)groups * 4 * 16 * kListK * kPTW * 4));
    CK(hipMalloc(&dStats, 16));
    CK(hipMemcpy(dWin, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dCur, cur.data(), cur.size() * 4, hipMemcpyHostToDevice));
    // ---- check: every candidate once (one workgroup, 5 batches of 4 x 64 >= 1089), sums against the host's tree
    {
        Params P{dWin, dCur, dCurRows, dSums, dLists, dStats, 5, 2};
        hipLaunchKernelGGL(batch_eval_kernel, dim3(1), dim3(256), 0, 0, P);
        CK(hipDeviceSynchronize());
        std::vector<float> sums((size_t)kCand * 16 * kPTW);
        CK(hipMemcpy(sums.data(), dSums, sums.size() * 4, hipMemcpyDeviceToHost));
        auto dist = [&](int col, int row, int dx, int dy) {
            const uint32_t p = win[(col + dx + kR) * kWinH + (row + dy + kR)], c = cur[col * kCP + row];
            float n = 0.0f;
            for (int k = 0; k < 4; ++k) { const float d = (float)((p >> (8 * k)) & 255u) - (float)((c >> (8 * k)) & 255u); n += d * d; }
            return sqrtf(n);
        };
        double worst = 0.0;
        for (int t = 0; t < 4000; ++t) {
            const int rank = rnd() % kCand, px = rnd() % kPTW, py = rnd() % 16;
            const int dy = rank / kSide - kR, dx = rank % kSide - kR;
            float V8[8];
            for (int i = 0; i < 8; ++i) {
                float d[8];
                for (int j = 0; j < 8; ++j) d[j] = dist(px + i, py + j, dx, dy);
                V8[i] = ((d[0] + d[1]) + (d[2] + d[3])) + ((d[4] + d[5]) + (d[6] + d[7]));
            }
            const float S = ((V8[0] + V8[1]) + (V8[2] + V8[3])) + ((V8[4] + V8[5]) + (V8[6] + V8[7]));
            const float got = sums[((size_t)rank * 16 + py) * kPTW + px];
            worst = fmax(worst, fabs((double)got - (double)S) / fmax(1.0, (double)S));
        }
        printf("check: 4000 sums against the host's tree, worst relative difference %.3g (v_sqrt_f32 is within 1 ulp)\n", worst);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {
        Params P{dWin, dCur, dCurRows, dSums, dLists, dStats, batches, mode};
        CK(hipMemset(dStats, 0, 16));
        hipLaunchKernelGGL(batch_eval_kernel, dim3(groups), dim3(256), 0, 0, P);      // warm-up
        CK(hipDeviceSynchronize());
        CK(hipMemset(dStats, 0, 16));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(batch_eval_kernel, dim3(groups), dim3(256), 0, 0, P);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t st[4];
        CK(hipMemcpy(st, dStats, 16, hipMemcpyDeviceToHost));
        const double cands = (double)groups * 4 * batches * 64;
        // 1,024 SIMDs; the staging of the window is included (one per workgroup)
        printf("mode %d (%s): %d workgroups x 4 waves x %d batches: %.3f ms, %.3f us per candidate and SIMD (the prefilter's evaluation: 0.71), "
               "passes per pixel and wave-batch %.3f, records %u\n",
               mode, mode ? "thresholds from +inf, shared by the four waves" : "nothing passes", groups, batches, ms,
               ms * 1e3 * 1024.0 / cands, (double)st[0] / ((double)groups * 4 * batches * 16 * kPTW), st[1]);
    }
    return 0;
}
