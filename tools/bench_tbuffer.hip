// Throughput of typed vs raw buffer loads on gfx950, L1/L2-resident window (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ inline i32x4 mk(const void *base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base; i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = __builtin_amdgcn_readfirstlane((int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (10u << 15)));
    return r;
}

template <int MODE>   // 0 typed xyzw, 1 raw dword, 2 typed batch of 10 then wait, 3 raw dwordx4 (16B/lane)
__global__ void k(const uint8_t *src, float *out, int iters, int window) {
    const i32x4 r = mk(src + (size_t)blockIdx.x % 64 * window, (uint32_t)window);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float acc = 0.f;
    int off = (wv * 284 * 4 + lane * 4) % (window - 64);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
            f32x4 p[10];
#pragma unroll
            for (int j = 0; j < 10; ++j)
                asm volatile("tbuffer_load_format_xyzw %0, %1, %2, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen"
                             : "=v"(p[j]) : "v"((off + j * 2048) % (window - 64)), "s"(r) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]) :: "memory");
#pragma unroll
            for (int j = 0; j < 10; ++j) acc += p[j].x + p[j].w;
        } else if (MODE == 1) {
            float p[10];
#pragma unroll
            for (int j = 0; j < 10; ++j)
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(p[j]) : "v"((off + j * 2048) % (window - 64)), "s"(r) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]) :: "memory");
#pragma unroll
            for (int j = 0; j < 10; ++j) acc += p[j];
        } else {
            f32x4 p[10];
#pragma unroll
            for (int j = 0; j < 10; ++j)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(p[j]) : "v"(((off * 4) + j * 2048) % (window - 1024)), "s"(r) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]), "+v"(p[7]), "+v"(p[8]), "+v"(p[9]) :: "memory");
#pragma unroll
            for (int j = 0; j < 10; ++j) acc += p[j].x + p[j].w;
        }
        off = (off + 4) % (window - 64);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F> double timeit(F f) {
    hipEvent_t b, e; hipEventCreate(&b); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    hipEventRecord(b); f(); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, b, e); return ms;
}

int main() {
    const int window = 48 * 1024;
    uint8_t *src; float *out;
    hipMalloc(&src, 64 * window + 4096); hipMemset(src, 7, 64 * window + 4096);
    hipMalloc(&out, 512 * 512 * 4);
    const int iters = 2000, blocks = 512;   // 512 threads per block: 2 blocks per CU -> 16 waves/CU
    const double loads = (double)blocks * 8 * iters * 10;   // wave-instructions
    double t;
    t = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, src, out, iters, window); });
    printf("tbuffer_load_format_xyzw (4B->16B/lane): %.3f ms, %.1f cycles per wave-load per CU @2.4GHz\n", t, t * 1e-3 * 2.4e9 / (loads / 256));
    t = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, src, out, iters, window); });
    printf("buffer_load_dword (4B/lane):             %.3f ms, %.1f cycles per wave-load per CU\n", t, t * 1e-3 * 2.4e9 / (loads / 256));
    t = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, src, out, iters, window); });
    printf("buffer_load_dwordx4 (16B/lane):          %.3f ms, %.1f cycles per wave-load per CU\n", t, t * 1e-3 * 2.4e9 / (loads / 256));
    return 0;
}
