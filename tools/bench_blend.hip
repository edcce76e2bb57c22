// What does the chip give for interpolate's traffic shape on content where it samples (read 2 x 33.2 MB of frames + 16.6 MB of
// vectors, write 33.2 MB), and which load form gets closest?  GPU box only; round 4.
//   xor      : two 16-byte loads per lane, out = p ^ c, one 16-byte store -- the memory floor of the shape
//   unpack   : the same loads, v_cvt_f32_ubyteN + the two-op UNORM conversion + mix + pack (the oracle's arithmetic)
//   fmt      : eight format-converting loads per lane (8_8_8_8 UNORM: the texture-address unit hands over byte / 255.0f), lanes own
//              four ADJACENT pixels (16-byte lane stride per load), mix + pack, one 16-byte store
//   fmtdense : the same loads with lane l on pixel 64 k + l (every load a dense 256 bytes), four 4-byte stores
// R = rows per thread (all loads of the R rows issued before the first use).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __device__ f32x4 lfg_llvm_raw_ptr_buffer_load_format_v4f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v4f32");
constexpr int kUnorm = (int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (0u << 12) | (10u << 15));
__device__ constexpr float kHi = 0x1.010102p-8f, kLo = -0x1.fdfdfep-33f;
__device__ __forceinline__ float un(float k) { return __builtin_fmaf(k, kHi, k * kLo); }
__device__ __forceinline__ float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }
__device__ __forceinline__ uint32_t pack(float r, float g, float b, float a) {
    uint32_t p = __builtin_amdgcn_cvt_pk_u8_f32(r * 255.0f, 0u, 0u);
    p = __builtin_amdgcn_cvt_pk_u8_f32(g * 255.0f, 1u, p);
    p = __builtin_amdgcn_cvt_pk_u8_f32(b * 255.0f, 2u, p);
    return __builtin_amdgcn_cvt_pk_u8_f32(a * 255.0f, 3u, p);
}
__device__ __forceinline__ uint32_t blend_bytes(uint32_t p, uint32_t c, float t) {
    return pack(mixf(un((float)(p & 0xff)), un((float)(c & 0xff)), t), mixf(un((float)((p >> 8) & 0xff)), un((float)((c >> 8) & 0xff)), t),
                mixf(un((float)((p >> 16) & 0xff)), un((float)((c >> 16) & 0xff)), t), mixf(un((float)(p >> 24)), un((float)(c >> 24)), t));
}

// MODE 0 xor, 1 unpack.  grid: (W/4/64, H/(4 R)); block 256 = 4 waves, wave = row group
template <int MODE, int R, bool MV>
__global__ __launch_bounds__(256) void k_wide(const uint8_t *__restrict__ prev, const uint8_t *__restrict__ curr, const uint8_t *__restrict__ mv,
                                              uint8_t *__restrict__ out, int W, int H, float t) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * R;
    if (qx * 4 >= W || py0 >= H) return;
    uint2 m[R];
    if (MV) {
#pragma unroll
        for (int r = 0; r < R; ++r) m[r] = *reinterpret_cast<const uint2 *>(mv + ((size_t)(py0 + r) * W + qx * 4) * 2);
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) any = any || (m[r].x | m[r].y) != 0u;
        if (any) return;
    }
    uint4 p[R], c[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const size_t off = ((size_t)(py0 + r) * W + qx * 4) * 4;
        p[r] = *reinterpret_cast<const uint4 *>(prev + off);
        c[r] = *reinterpret_cast<const uint4 *>(curr + off);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const size_t off = ((size_t)(py0 + r) * W + qx * 4) * 4;
        uint4 o;
        if (MODE == 0) o = uint4{p[r].x ^ c[r].x, p[r].y ^ c[r].y, p[r].z ^ c[r].z, p[r].w ^ c[r].w};
        else o = uint4{blend_bytes(p[r].x, c[r].x, t), blend_bytes(p[r].y, c[r].y, t), blend_bytes(p[r].z, c[r].z, t), blend_bytes(p[r].w, c[r].w, t)};
        *reinterpret_cast<uint4 *>(out + off) = o;
    }
}

// format loads.  DENSE = false: lane owns four adjacent pixels; true: lane l owns pixels 64 k + l of the wave's 256
template <bool DENSE, int R, bool MV>
__global__ __launch_bounds__(256) void k_fmt(const uint8_t *__restrict__ prev, const uint8_t *__restrict__ curr, const uint8_t *__restrict__ mv,
                                             uint8_t *__restrict__ out, int W, int H, float t) {
    const int lane = threadIdx.x & 63;
    const int qx = blockIdx.x * 64 + lane;
    const int py0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * R;
    if (qx * 4 >= W || py0 >= H) return;
    if (MV) {
        bool any = false;
#pragma unroll
        for (int r = 0; r < R; ++r) { const uint2 m = *reinterpret_cast<const uint2 *>(mv + ((size_t)(py0 + r) * W + qx * 4) * 2); any = any || (m.x | m.y) != 0u; }
        if (any) return;
    }
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(prev), 0, W * H * 4, kUnorm);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(curr), 0, W * H * 4, kUnorm);
    f32x4 p[R][4], c[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int base = ((py0 + r) * W + (DENSE ? blockIdx.x * 256 + lane : qx * 4)) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) p[r][k] = lfg_llvm_raw_ptr_buffer_load_format_v4f32(rP, base + (DENSE ? 256 : 4) * k, 0, 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) c[r][k] = lfg_llvm_raw_ptr_buffer_load_format_v4f32(rC, base + (DENSE ? 256 : 4) * k, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack(mixf(p[r][k].x, c[r][k].x, t), mixf(p[r][k].y, c[r][k].y, t), mixf(p[r][k].z, c[r][k].z, t), mixf(p[r][k].w, c[r][k].w, t));
        if (DENSE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<uint32_t *>(out + ((size_t)(py0 + r) * W + blockIdx.x * 256 + lane + 64 * k) * 4) = o[k];
        } else {
            *reinterpret_cast<uint4 *>(out + ((size_t)(py0 + r) * W + qx * 4) * 4) = uint4{o[0], o[1], o[2], o[3]};
        }
    }
}

template <typename F> void timeit(const char *name, F f, int reps = 200) {
    hipEvent_t b, e; hipEventCreate(&b); hipEventCreate(&e);
    for (int i = 0; i < 5; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(b);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, b, e);
    const double us = ms * 1e3 / reps;
    printf("%-28s %8.2f us per launch  (116.1 MB -> %.2f TB/s, %.3f of 8)\n", name, us, 116.1216e6 / us / 1e6, 116.1216e6 / us / 1e6 / 8.0);
}

int main() {
    const int W = 3840, H = 2160;
    const size_t n = (size_t)W * H * 4;
    uint8_t *prev, *curr, *out, *mv, *ref;
    hipMalloc(&prev, n); hipMalloc(&curr, n); hipMalloc(&out, n); hipMalloc(&ref, n); hipMalloc(&mv, n / 2);
    std::vector<uint8_t> h(n);
    uint32_t s = 12345u;
    for (auto &b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    hipMemcpy(prev, h.data(), n, hipMemcpyHostToDevice);
    for (auto &b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    hipMemcpy(curr, h.data(), n, hipMemcpyHostToDevice);
    hipMemset(mv, 0, n / 2);
    const float t = 0.5f;
#define GRID(R) dim3(W / 4 / 64, H / (4 * (R)))
    // results agree?
    hipLaunchKernelGGL((k_wide<1, 1, false>), GRID(1), dim3(256), 0, 0, prev, curr, mv, ref, W, H, t);
    auto check = [&](const char *name) {
        std::vector<uint8_t> a(n), b(n);
        hipMemcpy(a.data(), ref, n, hipMemcpyDeviceToHost); hipMemcpy(b.data(), out, n, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += a[i] != b[i];
        printf("%s vs unpack: %zu differing bytes\n", name, bad);
    };
    hipLaunchKernelGGL((k_fmt<false, 1, true>), GRID(1), dim3(256), 0, 0, prev, curr, mv, out, W, H, t); check("fmt");
    hipMemset(out, 0, n);
    hipLaunchKernelGGL((k_fmt<true, 2, true>), GRID(2), dim3(256), 0, 0, prev, curr, mv, out, W, H, t); check("fmtdense R2");
    hipMemset(out, 0, n);
    hipLaunchKernelGGL((k_wide<1, 4, true>), GRID(4), dim3(256), 0, 0, prev, curr, mv, out, W, H, t); check("unpack R4 mv");
#define T(name, K, R) timeit(name, [&] { hipLaunchKernelGGL(K, GRID(R), dim3(256), 0, 0, prev, curr, mv, out, W, H, t); })
    T("xor R1", (k_wide<0, 1, false>), 1); T("xor R2", (k_wide<0, 2, false>), 2); T("xor R4", (k_wide<0, 4, false>), 4);
    T("xor R1 +mv", (k_wide<0, 1, true>), 1); T("xor R2 +mv", (k_wide<0, 2, true>), 2); T("xor R4 +mv", (k_wide<0, 4, true>), 4);
    T("unpack R1", (k_wide<1, 1, false>), 1); T("unpack R2", (k_wide<1, 2, false>), 2); T("unpack R4", (k_wide<1, 4, false>), 4);
    T("unpack R1 +mv", (k_wide<1, 1, true>), 1); T("unpack R2 +mv", (k_wide<1, 2, true>), 2); T("unpack R4 +mv", (k_wide<1, 4, true>), 4);
    T("fmt R1", (k_fmt<false, 1, false>), 1); T("fmt R2", (k_fmt<false, 2, false>), 2);
    T("fmt R1 +mv", (k_fmt<false, 1, true>), 1); T("fmt R2 +mv", (k_fmt<false, 2, true>), 2);
    T("fmtdense R1", (k_fmt<true, 1, false>), 1); T("fmtdense R2", (k_fmt<true, 2, false>), 2);
    T("fmtdense R1 +mv", (k_fmt<true, 1, true>), 1); T("fmtdense R2 +mv", (k_fmt<true, 2, true>), 2);
    return 0;
}
