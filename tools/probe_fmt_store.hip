// GPU probe: buffer_store_format_xyzw through 8_8_8_8 descriptors -- what the store path converts a float to
// (rounding, clamping) for UNORM and USCALED, and how fast 4-byte-per-lane format stores stream compared with
// 16-byte plain stores.  Also buffer_load_format_xyzw USCALED for all 256 byte values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
extern "C" __device__ void st_fmt(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.store.format.v4f32");
extern "C" __device__ f32x4 ld_fmt(__amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v4f32");
constexpr int word3(int numfmt) { return (int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | ((unsigned)numfmt << 12) | (10u << 15)); }

template <int NF> __global__ void probe(const float *in, uint32_t *out, int n) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out, 0, n * 4, word3(NF));
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = in[i];
    st_fmt(f32x4{v, v, v, v}, r, i * 4, 0, 0);
}
__global__ void probe_load(const uint32_t *in, float *out, int n) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(in), 0, n * 4, word3(2));
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    f32x4 v = ld_fmt(r, i * 4, 0, 0);
    out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
}
// stream: each lane writes 4 adjacent pixels per row-iteration, as the scale kernel would
template <int MODE, int AUX> __global__ void stream(uint8_t *out, int W, int H, int pitch) {
    __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(out, 0, H * pitch, word3(2));
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(out, 0, H * pitch, 0x00020000);
    const int lanesPerRow = W / 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < lanesPerRow * H; i += gridDim.x * blockDim.x) {
        const int y = i / lanesPerRow, x4 = (i - y * lanesPerRow) * 4;
        const int off = y * pitch + x4 * 4;
        const float f = (float)(i & 255);
        if (MODE == 0) {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{(unsigned)i, (unsigned)i + 1, (unsigned)i + 2, (unsigned)i + 3}, rr, off, 0, AUX);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) st_fmt(f32x4{f, f + 1, f + 2, f + 3}, rf, off + 4 * k, 0, AUX);
        }
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
    std::vector<float> vals = {0.f, 0.4f, 0.5f, 0.6f, 1.0f, 1.5f, 2.5f, 3.5f, 254.5f, 255.f, 255.5f, 256.f, 300.f, -0.4f, -3.f, 1e9f, 100.49999f, 0.5f / 255.f, 1.5f / 255.f, 2.5f / 255.f, 1.0f, 1.1f, -0.1f, 0.25f, 127.5f / 255.f, 128.5f / 255.f};
    float *din; uint32_t *dout; hipMalloc(&din, vals.size() * 4); hipMalloc(&dout, vals.size() * 4);
    hipMemcpy(din, vals.data(), vals.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> o(vals.size());
    const char *names[] = {"UNORM", "SNORM", "USCALED", "SSCALED", "UINT", "SINT"};
    for (int nf : {0, 2, 4}) {
        hipMemset(dout, 0xEE, vals.size() * 4);
        if (nf == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(64), 0, 0, din, dout, (int)vals.size());
        if (nf == 2) hipLaunchKernelGGL(probe<2>, dim3(1), dim3(64), 0, 0, din, dout, (int)vals.size());
        if (nf == 4) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, din, dout, (int)vals.size());
        hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
        printf("store format 8_8_8_8 %s:", names[nf]);
        for (size_t i = 0; i < vals.size(); ++i) printf(" %g->%u", vals[i], o[i] & 0xff);
        printf("\n");
    }
    // load USCALED, all byte values
    std::vector<uint32_t> bytes(64); for (int i = 0; i < 64; ++i) bytes[i] = (4 * i) | ((4 * i + 1) << 8) | ((4 * i + 2) << 16) | ((unsigned)(4 * i + 3) << 24);
    uint32_t *db; float *df; hipMalloc(&db, 256); hipMalloc(&df, 1024);
    hipMemcpy(db, bytes.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_load, dim3(1), dim3(64), 0, 0, db, df, 64);
    std::vector<float> fl(256); hipMemcpy(fl.data(), df, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) bad += fl[i] != (float)i;
    printf("load format USCALED: %d of 256 byte values differ from (float)k\n", bad);
    // streaming
    const int W = 3840, H = 2160, pitch = W * 4;
    uint8_t *img; hipMalloc(&img, (size_t)H * pitch);
    for (int blocks : {1024, 2048}) {
        printf("blocks %d: plain b128 aux0 %.2f us, nt %.2f | format xyzw x4 aux0 %.2f, nt %.2f (event-timed, min of 20)\n", blocks,
               timeit([&] { hipLaunchKernelGGL((stream<0, 0>), dim3(blocks), dim3(256), 0, 0, img, W, H, pitch); }, 20),
               timeit([&] { hipLaunchKernelGGL((stream<0, 2>), dim3(blocks), dim3(256), 0, 0, img, W, H, pitch); }, 20),
               timeit([&] { hipLaunchKernelGGL((stream<1, 0>), dim3(blocks), dim3(256), 0, 0, img, W, H, pitch); }, 20),
               timeit([&] { hipLaunchKernelGGL((stream<1, 2>), dim3(blocks), dim3(256), 0, 0, img, W, H, pitch); }, 20));
    }
    return 0;
}
