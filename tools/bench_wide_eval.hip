// Throughput probe: the prefilter's full evaluation in its own form (csrc/prefilter_sums.inc: fetchWindow / columnSums / transpose /
// runSums + the packed "does anything pass" test; lane = position column, one candidate per pass, software-pipelined as
// in the kernel) at TWO and at THREE waves per SIMD.  The shipped kernel holds 256 VGPRs and 77 KB of LDS per workgroup: two
// workgroups per CU.  This loop alone -- no lattice tests, no record path, no narrow search -- is what a search-only kernel
// for handed-over segments would run; the question is whether it fits 168 VGPRs, and what the third wave buys.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -DOCC=2|3 -o bench_wide_eval_occN bench_wide_eval.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#ifndef OCC
#define OCC 2
#endif

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) uint32_t *lds_ro_u32_ptr;
typedef const __attribute__((address_space(3))) f32x2 *lds_ro_f32x2_ptr;

constexpr int kB = 8, kR = 16, kSide = 2 * kR + 1, kCand = kSide * kSide;
constexpr int kPTW = 56, kSeg = 16, kSegD = kSeg + kB - 1;
constexpr int kWinW = 95, kWinRows = kSegD + 2 * kR, kWinH = kWinRows | 1;      // a segment's 55 rows, column-major, odd pitch
constexpr int kSlabP = 132, kRun = 7, kRunIn = kRun + kB - 1;

__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

__global__ __launch_bounds__(256, OCC) void wide_eval_kernel(const uint32_t *__restrict__ win, const uint32_t *__restrict__ cur, float *out, int evals) {
    __shared__ uint32_t sWin[kWinW * kWinH];                           // 20.9 KB
    __shared__ __attribute__((aligned(8))) float sSlab[4][8 * kSlabP]; // 4 x 4.2 KB
#if OCC == 2
    __shared__ uint32_t sPad[10240];                                   // 40 KB: two workgroups per CU, as the shipped kernel's 77 KB allow
    if (evals < 0) sPad[threadIdx.x] = 1u;
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < kWinW * kWinH; i += 256) sWin[i] = win[i % (kWinW * 55)];
    __syncthreads();
    uint32_t c[kSegD], cc[kSegD];
#pragma unroll
    for (int j = 0; j < kSegD; ++j) { c[j] = cur[(lane * kSegD + j) % 1449]; cc[j] = __builtin_amdgcn_udot4(c[j], c[j], 0x4B000000u, false); }
    const lds_ro_u32_ptr winBase = (lds_ro_u32_ptr)(sWin + min(lane, kPTW + kB - 2) * kWinH);
    const int r8 = lane & 7, q = lane >> 3;
    f32x2 *const slabW = reinterpret_cast<f32x2 *>(sSlab[wave]) + lane;
    const lds_ro_f32x2_ptr slabR = (lds_ro_f32x2_ptr)(sSlab[wave]) + r8 * (kSlabP / 2) + kRun * q;
    f32x2 thr2[kRun];
#pragma unroll
    for (int i = 0; i < kRun; ++i) thr2[i] = f32x2{-1.0f, -1.0f};     // nothing passes
    uint32_t p[kSegD];
    f32x2 x[kRunIn];
    float acc = 0.0f;
    auto fetch = [&](int e) {
        const int rank = (e * 4 + wave) % kCand;
        const lds_ro_u32_ptr w = winBase + (rank % kSide) * kWinH + rank / kSide;
#pragma unroll
        for (int j = 0; j < kSegD; ++j) p[j] = w[j];
    };
    bool pending = false;
    for (int e = 0; e <= evals; ++e) {
        const bool have = e < evals;
        if (have) fetch(e);
        if (pending) {
            f32x2 h2[kRunIn - 1], h4[kRunIn - 3], s2[kRun];
#pragma unroll
            for (int i = 0; i < kRunIn - 1; ++i) h2[i] = x[i] + x[i + 1];
#pragma unroll
            for (int i = 0; i < kRunIn - 3; ++i) h4[i] = h2[i] + h2[i + 2];
#pragma unroll
            for (int i = 0; i < kRun; ++i) s2[i] = h4[i] + h4[i + 4];
            f32x2 dm = thr2[0] - s2[0];
            float top = __builtin_fmaxf(dm.x, dm.y);
#pragma unroll
            for (int i = 1; i < kRun; ++i) { dm = thr2[i] - s2[i]; top = __builtin_fmaxf(top, __builtin_fmaxf(dm.x, dm.y)); }
            if (__builtin_amdgcn_readfirstlane(__ballot(top >= 0.0f) != 0ull)) acc += top;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (have) {
            auto f1of = [&](int j) { return __builtin_bit_cast(float, __builtin_amdgcn_udot4(p[j], p[j], cc[j], false)); };
            auto f2of = [&](int j) { return __builtin_bit_cast(float, __builtin_amdgcn_udot4(c[j], p[j], 0x4B800000u, false)); };
            const f32x2 kBias = {8388608.0f, 8388608.0f};
            constexpr int kPairs = kSegD - 8;
            f32x2 A[kPairs];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x2 N = (f32x2{f1of(j), f1of(j + 8)} - f32x2{f2of(j), f2of(j + 8)}) + kBias;
                A[j] = f32x2{__builtin_amdgcn_sqrtf(N.x), __builtin_amdgcn_sqrtf(N.y)};
            }
            float dHi[kSegD - 16];
#pragma unroll
            for (int k = 0; k + 1 < kSegD - 16; k += 2) {
                const f32x2 N = (f32x2{f1of(16 + k), f1of(17 + k)} - f32x2{f2of(16 + k), f2of(17 + k)}) + kBias;
                dHi[k] = __builtin_amdgcn_sqrtf(N.x); dHi[k + 1] = __builtin_amdgcn_sqrtf(N.y);
            }
            dHi[kSegD - 17] = __builtin_amdgcn_sqrtf((f1of(kSegD - 1) - f2of(kSegD - 1)) + 8388608.0f);
#pragma unroll
            for (int j = 8; j < kPairs; ++j) A[j] = f32x2{A[j - 8].y, dHi[j - 8]};
            f32x2 Bt[kPairs - 1], G[kPairs - 3], C8[8];
#pragma unroll
            for (int j = 0; j < kPairs - 1; ++j) Bt[j] = A[j] + A[j + 1];
#pragma unroll
            for (int j = 0; j < kPairs - 3; ++j) G[j] = Bt[j] + Bt[j + 2];
#pragma unroll
            for (int j = 0; j < 8; ++j) C8[j] = G[j] + G[j + 4];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 8; ++r) slabW[r * (kSlabP / 2)] = C8[r];
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < kRunIn; ++i) x[i] = slabR[i];
            wave_lds_sync();
        }
        pending = have;
    }
    out[blockIdx.x * 256 + tid] = acc + x[0].x;
#if OCC == 2
    if (evals < 0) out[0] = (float)sPad[255 - threadIdx.x];
#endif
}

int main(int argc, char **argv) {
    const int evals = argc > 1 ? atoi(argv[1]) : 400;
    std::vector<uint32_t> win(kWinW * 55), cur(1449);
    uint32_t s = 777u;
    for (auto &v : win) { s = s * 1664525u + 1013904223u; v = s; }
    for (auto &v : cur) { s = s * 1664525u + 1013904223u; v = s; }
    uint32_t *dWin, *dCur; float *dOut;
    const int groups = 256 * OCC * 2;                                  // twice what the chip holds at once
    CK(hipMalloc(&dWin, win.size() * 4)); CK(hipMalloc(&dCur, cur.size() * 4)); CK(hipMalloc(&dOut, (size_t)groups * 256 * 4));
    CK(hipMemcpy(dWin, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dCur, cur.data(), cur.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int resident = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, wide_eval_kernel, 256, 0));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(wide_eval_kernel, dim3(groups), dim3(256), 0, 0, dWin, dCur, dOut, evals);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0.0f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("launch bound %d waves per SIMD, %d workgroups resident per CU: %d workgroups x 4 waves x %d evaluations in %.3f ms = %.3f us per evaluation and SIMD "
                        "(the shipped kernel, all in, on frames without a match: 0.71)\n", OCC, resident, groups, evals, ms, ms * 1e3 * 1024.0 / ((double)groups * 4 * evals));
    }
    return 0;
}
