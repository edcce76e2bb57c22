// What does the memory system give for scale's traffic shape (read 8.3 MB, write 33.2 MB)?  GPU box only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int AUX, int BYTES>   // each lane stores BYTES (8 or 16) per iteration, grid-stride
__global__ void fill(uint8_t *out, size_t n, unsigned v) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)n, 0x00020000);
    const size_t stride = (size_t)gridDim.x * blockDim.x * BYTES;
    for (size_t off = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * BYTES; off < n; off += stride) {
        if (BYTES == 16) __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v, v, v}, r, (int)off, 0, AUX);
        else __builtin_amdgcn_raw_buffer_store_b64(u32x2{v, v}, r, (int)off, 0, AUX);
    }
}

template <int AUX>
__global__ void expand(const uint32_t *in, uint8_t *out, size_t nIn) {   // read 4 B, write 16 B per lane
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)(nIn * 16), 0x00020000);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nIn; i += stride) {
        const unsigned v = in[i];
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{v, v + 1, v + 2, v + 3}, r, (int)(i * 16), 0, AUX);
    }
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t b, e; hipEventCreate(&b); hipEventCreate(&e);
    f(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int i = 0; i < reps; ++i) { hipEventRecord(b); f(); hipEventRecord(e); hipEventSynchronize(e); float ms; hipEventElapsedTime(&ms, b, e); if (ms < best) best = ms; }
    return best * 1e3f;
}

int main() {
    const size_t nOut = 3840ull * 2160 * 4, nIn = 1920ull * 1080;
    uint8_t *out; uint32_t *in;
    hipMalloc(&out, nOut); hipMalloc(&in, nIn * 4); hipMemset(in, 1, nIn * 4);
    for (int blocks : {1024, 2048, 4096}) {
        printf("blocks %d: fill16 plain %.2f us, nt %.2f, sc1 %.2f, nt+sc1 %.2f | fill8 plain %.2f, nt+sc1 %.2f | expand plain %.2f nt+sc1 %.2f\n", blocks,
               timeit([&] { hipLaunchKernelGGL((fill<0, 16>), dim3(blocks), dim3(256), 0, 0, out, nOut, 7u); }, 20),
               timeit([&] { hipLaunchKernelGGL((fill<2, 16>), dim3(blocks), dim3(256), 0, 0, out, nOut, 7u); }, 20),
               timeit([&] { hipLaunchKernelGGL((fill<16, 16>), dim3(blocks), dim3(256), 0, 0, out, nOut, 7u); }, 20),
               timeit([&] { hipLaunchKernelGGL((fill<18, 16>), dim3(blocks), dim3(256), 0, 0, out, nOut, 7u); }, 20),
               timeit([&] { hipLaunchKernelGGL((fill<0, 8>), dim3(blocks), dim3(256), 0, 0, out, nOut, 7u); }, 20),
               timeit([&] { hipLaunchKernelGGL((fill<18, 8>), dim3(blocks), dim3(256), 0, 0, out, nOut, 7u); }, 20),
               timeit([&] { hipLaunchKernelGGL((expand<0>), dim3(blocks), dim3(256), 0, 0, in, out, nIn); }, 20),
               timeit([&] { hipLaunchKernelGGL((expand<18>), dim3(blocks), dim3(256), 0, 0, in, out, nIn); }, 20));
    }
    printf("empty-ish kernel: %.2f us\n", timeit([&] { hipLaunchKernelGGL((fill<0, 16>), dim3(1), dim3(64), 0, 0, out, (size_t)1024, 7u); }, 20));
    return 0;
}
