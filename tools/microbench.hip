// Microbenchmarks for gfx950 facts the kernels' design depends on (run on the GPU box):
//   1. v_add_f32 vs v_pk_add_f32 throughput at 1/2/4 waves per SIMD
//   2. rounding / saturation behaviour of v_cvt_pk_u8_f32
//   3. cost of the correctly rounded sqrtf sequence vs v_sqrt_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int ITERS>
__global__ void add_chain(float *out, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed * (float)(i + threadIdx.x);
    float inc = seed;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(inc));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ITERS>
__global__ void pk_add_chain(float *out, float seed) {
    float2v a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i].x = seed * (float)(i + threadIdx.x); a[i].y = seed * (float)(i + 1 + threadIdx.x); }
    float2v inc = {seed, seed};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(inc));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ITERS>
__global__ void fma_chain(float *out, float seed) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed * (float)(i + threadIdx.x);
    float m = 1.0f + seed * 1e-6f;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(seed));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ITERS>
__global__ void pk_fma_chain(float *out, float seed) {
    float2v a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i].x = seed * (float)(i + threadIdx.x); a[i].y = seed * (float)(i + 1 + threadIdx.x); }
    float2v m = {1.0f + seed * 1e-6f, 1.0f + seed * 1e-6f};
    float2v c = {seed, seed};
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ITERS, bool EXACT>
__global__ void sqrt_chain(float *out, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 1.0f + seed * (float)(i + threadIdx.x);
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (EXACT) a[i] = __builtin_sqrtf(a[i] + 1.5f);
                else a[i] = __builtin_amdgcn_sqrtf(a[i] + 1.5f);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void cvt_probe(const float *in, unsigned *out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0u);
}

template <typename K>
static double time_kernel(K launch, int reps) {
    hipEvent_t b, e;
    hipEventCreate(&b); hipEventCreate(&e);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(b);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0;
    hipEventElapsedTime(&ms, b, e);
    return ms / reps;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    float *out;
    CK(hipMalloc(&out, 256 * 16 * 1024 * sizeof(float)));
    constexpr int IT = 2000;
    const int cus = prop.multiProcessorCount;
    for (int wavesPerSimd : {1, 2, 4}) {
        const int blocks = cus * wavesPerSimd;       // 256 threads = 4 waves = 1 wave per SIMD per block
        const double ops = (double)blocks * 256 * IT * 8 * 16;
        double t = time_kernel([&] { hipLaunchKernelGGL(add_chain<IT>, dim3(blocks), dim3(256), 0, 0, out, 1e-3f); }, 5);
        printf("v_add_f32     waves/SIMD %d: %.3f ms  %.2f T lane-adds/s\n", wavesPerSimd, t, ops / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(pk_add_chain<IT>, dim3(blocks), dim3(256), 0, 0, out, 1e-3f); }, 5);
        printf("v_pk_add_f32  waves/SIMD %d: %.3f ms  %.2f T lane-adds/s\n", wavesPerSimd, t, ops / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(fma_chain<IT>, dim3(blocks), dim3(256), 0, 0, out, 1e-3f); }, 5);
        printf("v_fma_f32     waves/SIMD %d: %.3f ms  %.2f T lane-fmas/s\n", wavesPerSimd, t, ops / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL(pk_fma_chain<IT>, dim3(blocks), dim3(256), 0, 0, out, 1e-3f); }, 5);
        printf("v_pk_fma_f32  waves/SIMD %d: %.3f ms  %.2f T lane-fmas/s\n", wavesPerSimd, t, ops / t / 1e9);
        const double sops = (double)blocks * 256 * 500 * 8 * 8;
        t = time_kernel([&] { hipLaunchKernelGGL((sqrt_chain<500, true>), dim3(blocks), dim3(256), 0, 0, out, 1e-3f); }, 5);
        printf("sqrtf exact+add waves/SIMD %d: %.3f ms  %.2f T/s\n", wavesPerSimd, t, sops / t / 1e9);
        t = time_kernel([&] { hipLaunchKernelGGL((sqrt_chain<500, false>), dim3(blocks), dim3(256), 0, 0, out, 1e-3f); }, 5);
        printf("v_sqrt_f32+add  waves/SIMD %d: %.3f ms  %.2f T/s\n", wavesPerSimd, t, sops / t / 1e9);
    }
    // cvt_pk_u8_f32 semantics
    std::vector<float> probe = {0.0f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 254.5f, 254.6f, 255.0f, 255.4f, 255.5f, 256.0f, 300.0f, -0.4f, -0.6f, -3.0f, 1e9f, NAN};
    float *din; unsigned *dout;
    CK(hipMalloc(&din, probe.size() * 4)); CK(hipMalloc(&dout, probe.size() * 4));
    CK(hipMemcpy(din, probe.data(), probe.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, din, dout, (int)probe.size());
    std::vector<unsigned> res(probe.size());
    CK(hipMemcpy(res.data(), dout, probe.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < probe.size(); ++i) printf("cvt_pk_u8_f32(%g) = %u\n", probe[i], res[i]);
    return 0;
}
