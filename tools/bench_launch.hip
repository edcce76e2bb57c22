// Launch cost of a grid whose workgroups exit at once, as a function of the kernel's register and LDS footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LDSF, int NV>
__global__ __launch_bounds__(512, 4) void k(const unsigned *flags, float *out) {
    __shared__ float s[LDSF];
    if (flags && flags[blockIdx.y * gridDim.x + blockIdx.x] == 0u) return;
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = out[threadIdx.x + i * 512];
    s[threadIdx.x] = threadIdx.x; __syncthreads();
    float acc = s[(threadIdx.x + 1) % 512];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc += v[i] * v[(i + 7) % NV];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}
template <typename F> float timeit(F f) {
    hipEvent_t b, e; hipEventCreate(&b); hipEventCreate(&e);
    f(); hipDeviceSynchronize(); float best = 1e9;
    for (int i = 0; i < 5; ++i) { hipEventRecord(b); f(); hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, b, e); if (ms < best) best = ms; }
    return best * 1e3f;
}
int main() {
    unsigned *flags; float *out; hipMalloc(&flags, 2040 * 4); hipMemset(flags, 0, 2040 * 4); hipMalloc(&out, 64 << 20);
    printf("lds 2KB  few vgpr : %.1f us\n", timeit([&] { hipLaunchKernelGGL((k<512, 4>), dim3(60, 34), dim3(512), 0, 0, flags, out); }));
    printf("lds 42KB few vgpr : %.1f us\n", timeit([&] { hipLaunchKernelGGL((k<10624, 4>), dim3(60, 34), dim3(512), 0, 0, flags, out); }));
    printf("lds 42KB ~100 vgpr: %.1f us\n", timeit([&] { hipLaunchKernelGGL((k<10624, 96>), dim3(60, 34), dim3(512), 0, 0, flags, out); }));
    printf("lds 78KB ~100 vgpr: %.1f us\n", timeit([&] { hipLaunchKernelGGL((k<19968, 96>), dim3(60, 34), dim3(512), 0, 0, flags, out); }));
    return 0;
}
