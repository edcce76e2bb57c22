// Probe: does the gfx950 texture-address unit convert RGBA8 UNORM -> float exactly like (float)k / 255.0f?
// Loads 64 packed RGBA8 texels (all 256 byte values) with tbuffer_load_format_xyzw / buffer_load_format_xyzw.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

typedef float float4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

__global__ void probe(const uint32_t *src, float4v *outT, float4v *outB, int nbytes) {
    // V# : base, stride 0, num_records = nbytes, word3 = DST_SEL xyzw = R,G,B,A ; NUM_FORMAT UNORM ; DATA_FORMAT 8_8_8_8
    const uint64_t base = (uint64_t)src;
    int4v rsrc;
    rsrc.x = (int)(uint32_t)base;
    rsrc.y = (int)((uint32_t)(base >> 32) & 0xffffu);
    rsrc.z = nbytes;
    rsrc.w = (int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (0u << 12) | (10u << 15));
    // make uniform
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
    rsrc.z = __builtin_amdgcn_readfirstlane(rsrc.z); rsrc.w = __builtin_amdgcn_readfirstlane(rsrc.w);
    const int off = threadIdx.x * 4;
    float4v a, b;
    asm volatile("tbuffer_load_format_xyzw %0, %2, %3, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
                 "buffer_load_format_xyzw %1, %2, %3, 0 offen\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(off), "s"(rsrc) : "memory");
    outT[threadIdx.x] = a;
    outB[threadIdx.x] = b;
}

int main() {
    std::vector<uint32_t> h(64);
    for (int i = 0; i < 64; ++i) h[i] = (uint32_t)(4 * i) | ((uint32_t)(4 * i + 1) << 8) | ((uint32_t)(4 * i + 2) << 16) | ((uint32_t)(4 * i + 3) << 24);
    uint32_t *d; float4v *oT, *oB;
    hipMalloc(&d, 256); hipMalloc(&oT, 64 * 16); hipMalloc(&oB, 64 * 16);
    hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, oT, oB, 256);
    std::vector<float> rT(256), rB(256);
    hipMemcpy(rT.data(), oT, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(rB.data(), oB, 1024, hipMemcpyDeviceToHost);
    int badT = 0, badB = 0;
    for (int k = 0; k < 256; ++k) {
        const float want = (float)k / 255.0f;
        if (memcmp(&want, &rT[k], 4)) { if (badT < 8) printf("tbuffer k=%d got %.9g want %.9g\n", k, rT[k], want); ++badT; }
        if (memcmp(&want, &rB[k], 4)) { if (badB < 8) printf("buffer  k=%d got %.9g want %.9g\n", k, rB[k], want); ++badB; }
    }
    printf("tbuffer_load_format_xyzw mismatches: %d / 256\nbuffer_load_format_xyzw mismatches: %d / 256\n", badT, badB);
    return 0;
}
