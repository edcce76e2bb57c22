// Does a CU mask on the stream of a chip-filling persistent kernel leave room for a kernel of RCCL's footprint?
// hog: 256 threads, 256 vector registers, 75 KB of LDS (two per CU: the persistent prefilter kernel's footprint), spins for 3 ms;
// probe: 256 threads, 261 registers (17 accumulation registers), 19,744 bytes of LDS, 8 workgroups of 50 us, on a high-priority stream.
//   hipcc --offload-arch=gfx950 -O2 -o build_variants/probe_cu_reserve tools/probe_cu_reserve.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <map>
#include <algorithm>
#include <unistd.h>

__global__ __launch_bounds__(256, 1) void hog_kernel(int ticks, uint32_t *sink, unsigned long long *where) {
    extern __shared__ uint32_t lds[];
    lds[threadIdx.x] = threadIdx.x;
    if (threadIdx.x == 0 && where) {
        where[blockIdx.x * 2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);
        where[blockIdx.x * 2 + 1] = wall_clock64();
    }
    asm volatile("v_mov_b32 v255, 0" ::: "v255");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
    __syncthreads();
    if (sink && lds[(threadIdx.x + 1) & 255] == 0xffffffffu) *sink = 1;
}

__global__ __launch_bounds__(256) void probe_kernel(int ticks, uint32_t *sink) {
    __shared__ uint32_t lds[19744 / 4];
    lds[threadIdx.x] = threadIdx.x;
    asm volatile("v_mov_b32 v243, 0\n\tv_accvgpr_write_b32 a16, 0" ::: "v243", "a16");
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (sink && lds[(threadIdx.x + 1) & 255] == 0xffffffffu) *sink = 1;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, words = (cus + 31) / 32;
    int least, greatest; hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t hiPrio; hipStreamCreateWithPriority(&hiPrio, hipStreamNonBlocking, greatest);
    hipFuncSetAttribute((const void *)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 75 * 1024);
    int perCu = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, hog_kernel, 256, 75 * 1024);
    printf("%d CUs, hog workgroups per CU %d\n", cus, perCu);
    for (int reserve : {0, 8, 16, 108, 116}) {
        for (int rep = 0; rep < 2; ++rep) {
            // (100 + n: the probe's stream is masked to the n reserved CUs alone, instead of a high-priority stream that may use any CU)
            const bool probeMasked = reserve >= 100;
            if (probeMasked) reserve -= 100;
            hipStream_t own = nullptr;
            if (probeMasked) {
                std::vector<uint32_t> only(words, 0u);
                for (int i = 0; i < reserve; ++i) only[i / 32] |= 1u << (i % 32);
                hipExtStreamCreateWithCUMask(&own, words, only.data());
            }
            hipStream_t hi = probeMasked ? own : hiPrio;
            std::vector<uint32_t> mask(words, 0xffffffffu);
            for (int i = 0; i < reserve; ++i) mask[i / 32] &= ~(1u << (i % 32));
            hipStream_t s;
            if (reserve) hipExtStreamCreateWithCUMask(&s, words, mask.data()); else hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            hipEvent_t a, b, c, d; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c); hipEventCreate(&d);
            const int grid = (cus - reserve) * perCu;
            hipEventRecord(c, s);
            unsigned long long *where; hipMalloc(&where, grid * 16); hipMemset(where, 0, grid * 16);
            hipLaunchKernelGGL(hog_kernel, dim3(grid), dim3(256), 75 * 1024, s, 300000, (uint32_t *)nullptr, where);       // 3 ms
            hipEventRecord(d, s);
            hipEventQuery(d);                                    // (flushes the stream)
            usleep(500);                                         // the probe half a millisecond later, while the hog is resident
            hipEventRecord(a, hi);
            hipLaunchKernelGGL(probe_kernel, dim3(8), dim3(256), 0, hi, 5000, (uint32_t *)nullptr);                  // 50 us
            hipEventRecord(b, hi);
            hipDeviceSynchronize();
            float probe = 0, hog = 0, since = 0; hipEventElapsedTime(&probe, a, b); hipEventElapsedTime(&hog, c, d); hipEventElapsedTime(&since, c, b);
            printf("%s reserve %2d  grid %3d  hog %.3f ms  probe(8 x 50 us) issued -> done %.3f ms, done %.3f ms after the hog's start\n", probeMasked ? "probe on the reserved CUs only:" : "probe on a high-priority stream:", reserve, grid, hog, probe, since);
            {
                std::vector<unsigned long long> h(grid * 2); hipMemcpy(h.data(), where, grid * 16, hipMemcpyDeviceToHost);
                unsigned long long t0 = ~0ull; for (int i = 0; i < grid; ++i) t0 = std::min(t0, h[i * 2 + 1]);
                std::map<int, int> perCuFirst, perCuAll; int first = 0, onReserved = 0;
                for (int i = 0; i < grid; ++i) {
                    const uint32_t hw = (uint32_t)h[i * 2]; const int xcc = (int)(h[i * 2] >> 32) & 15;
                    const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
                    const int key = xcc * 1000 + (se * 2 + sh) * 16 + cu;
                    const bool early = h[i * 2 + 1] - t0 < 100000;     // first millisecond
                    if (early) { ++first; ++perCuFirst[key]; }
                    ++perCuAll[key];
                    if (se * 2 + sh == 0 && cu == 0) ++onReserved;
                }
                int two = 0; for (auto &q : perCuFirst) if (q.second >= 2) ++two;
                printf("      hog: %d of %d workgroups started in the first ms, on %zu CUs (%d of them with two); CUs used over all %zu; workgroups on (se0 cu0) of any XCD %d\n",
                       first, grid, perCuFirst.size(), two, perCuAll.size(), onReserved);
            }
            hipFree(where);
            hipStreamDestroy(s);
            if (own) hipStreamDestroy(own);
            if (probeMasked) reserve += 100;
        }
    }
    return 0;
}
