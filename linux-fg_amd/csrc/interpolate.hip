// interpolate.hip -- motion-displaced blend, the MI355X-native replacement of
// shaders/interpolate.comp (reference: /root/reference/shaders/interpolate.comp:15-40, dispatched by
// src/frame_manager.cpp:351-366).
//
// Literal reference semantics: the motion vector is in whole pixels and is added to
// the NORMALISED uv (interpolate.comp:16), so a non-zero component shifts the sample by whole image
// fractions or pushes it out of [0,1], where that source contributes vec4(0) (:17-20).
//
// The arithmetic repeats the oracle's operation order exactly (the library is built with
// -ffp-contract=off): uv division, bilinear weights per the Vulkan rules with fp32 fractions,
// mix(x,y,a) = x*(1-a) + y*a, clamp, *255, round half to even.  UNORM8 -> float uses the exact
// two-op form in lfg_device.hpp.  Output therefore matches the oracle bit for bit.
//
// Roofline: HBM.  Algorithmic bytes per pixel = 4 (prev) + 4 (curr) + 2 (mv) + 4 (out) = 14.
// One thread produces four horizontally adjacent pixels: one 8-byte MV load, one 16-byte store.
#include "lfg_device.hpp"
#include "lfg_internal.hpp"
#include "lfg_interp.hpp"

namespace lfg {

// INTENDED = false: interpolate.comp as written.  INTENDED = true (opt-in, lfg_set_semantics):
// the motion vector is divided by the image size before it is added to uv, so it displaces by pixels.
template <bool INTENDED>
__global__ __launch_bounds__(256) void interpolate_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    const int8_t *__restrict__ mv, int mvPitch, uint8_t *__restrict__ out, int outPitch,
    int W, int H, float t, const float *__restrict__ uvxTable, const float *__restrict__ uvyTable) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);         // group of four pixels
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int px0 = qx * 4;
    if (px0 >= W || py >= H) return;
    // ((float)p + 0.5f) / (float)size, interpolate.comp:30, from the host's tables (lfg_internal.hpp: UvTable; the tables are
    // padded to whole groups of four)
    const float uvy = uvyTable[py];
    const float4 uvx4 = *reinterpret_cast<const float4 *>(uvxTable + px0);
    const float uvxOf[4] = {uvx4.x, uvx4.y, uvx4.z, uvx4.w};
    const int8_t *mrow = mv + (size_t)py * (size_t)mvPitch;
    uint8_t *orow = out + (size_t)py * (size_t)outPitch;
    const bool full = (px0 + 3 < W) && ((mvPitch & 7) == 0) && ((outPitch & 15) == 0);

    int8_t m[8];
#ifdef LFG_DIAG_MV_UNIFORM             // (timing experiment, DESIGN.md 4.4: what reading the vectors costs -- every pixel takes the
                                       //  vector of the frame's centre, one scalar load; right for the benchmark's pan but for its rim)
    if (true) {
        const int8_t *centre = mv + (size_t)(H / 2) * (size_t)mvPitch + (size_t)(W / 2) * 2u;
        const int8_t cx = centre[0], cy = centre[1];
#pragma unroll
        for (int i = 0; i < 4; ++i) { m[2 * i] = cx; m[2 * i + 1] = cy; }
    } else
#endif
    if (full) {
        const uint2 mm = *reinterpret_cast<const uint2 *>(mrow + (size_t)px0 * 2u);
        m[0] = (int8_t)(mm.x & 0xff); m[1] = (int8_t)((mm.x >> 8) & 0xff);
        m[2] = (int8_t)((mm.x >> 16) & 0xff); m[3] = (int8_t)(mm.x >> 24);
        m[4] = (int8_t)(mm.y & 0xff); m[5] = (int8_t)((mm.y >> 8) & 0xff);
        m[6] = (int8_t)((mm.y >> 16) & 0xff); m[7] = (int8_t)(mm.y >> 24);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = min(px0 + i, W - 1);
            m[2 * i] = mrow[(size_t)px * 2u]; m[2 * i + 1] = mrow[(size_t)px * 2u + 1];
        }
    }

    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uvx = uvxOf[i];
        float mx = (float)m[2 * i], my = (float)m[2 * i + 1];
        if (INTENDED) { mx = mx / (float)W; my = my / (float)H; }
        const V4 p = sample_with_motion(prev, W, H, prevPitch, uvx, uvy, mx, my, -t);
        const V4 c = sample_with_motion(curr, W, H, currPitch, uvx, uvy, mx, my, 1.0f - t);
        o[i] = pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
    }
    if (full) {
        *reinterpret_cast<uint4 *>(orow + (size_t)px0 * 4u) = uint4{o[0], o[1], o[2], o[3]};
    } else {
        for (int i = 0; i < 4 && px0 + i < W; ++i)
            *reinterpret_cast<uint32_t *>(orow + (size_t)(px0 + i) * 4u) = o[i];
    }
}

hipError_t launch_interpolate(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                              const lfg_frame &mv, const lfg_frame &out, float factor, bool intended, const float *uvx, const float *uvy) {
    const int quads = ((int)out.width + 3) / 4;
    dim3 grid((quads + 63) / 64, (out.height + 3) / 4);
    if (intended)
        hipLaunchKernelGGL(interpolate_kernel<true>, grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, (uint8_t *)out.data, (int)out.pitch,
                           (int)out.width, (int)out.height, factor, uvx, uvy);
    else
        hipLaunchKernelGGL(interpolate_kernel<false>, grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, (uint8_t *)out.data, (int)out.pitch,
                           (int)out.width, (int)out.height, factor, uvx, uvy);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ several factors per pair
//
// BASELINE config 5 (60 -> 240 fps) generates three frames per pair, t = 1/4, 1/2, 3/4, from ONE motion field.  Run
// as three dispatches of the kernel above, prev, curr and the vectors are read three times: 3 x 14 = 42 B/pixel.
// Here one thread reads its four vectors once and produces the pixels of all N frames: 10 + 4N B/pixel (22 for N = 3).
// Where the vector is (0,0) -- static content, and under the literal semantics the only case in which both sources
// are sampled inside the image for every t -- the two texels are fetched once and only mix() is repeated.
// Every output byte is computed by the same functions, in the same order, as interpolate_kernel computes it: the
// frames are identical to N separate calls (tests/test_gpu_parity.py).
constexpr int kMaxMulti = 4;
struct MultiTargets {
    uint8_t *out[kMaxMulti];
    int pitch[kMaxMulti];
    float t[kMaxMulti];
};

template <int N, bool INTENDED>
__global__ __launch_bounds__(256) void interpolate_multi_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    const int8_t *__restrict__ mv, int mvPitch, MultiTargets tg, int W, int H, bool wideStores,
    const float *__restrict__ uvxTable, const float *__restrict__ uvyTable) {
    const int qx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int px0 = qx * 4;
    if (px0 >= W || py >= H) return;
    const float uvy = uvyTable[py];
    const float4 uvx4 = *reinterpret_cast<const float4 *>(uvxTable + px0);
    const float uvxOf[4] = {uvx4.x, uvx4.y, uvx4.z, uvx4.w};
    const int8_t *mrow = mv + (size_t)py * (size_t)mvPitch;
    const bool full = (px0 + 3 < W) && ((mvPitch & 7) == 0) && wideStores;

    int8_t m[8];
    if (full) {
        const uint2 mm = *reinterpret_cast<const uint2 *>(mrow + (size_t)px0 * 2u);
        m[0] = (int8_t)(mm.x & 0xff); m[1] = (int8_t)((mm.x >> 8) & 0xff);
        m[2] = (int8_t)((mm.x >> 16) & 0xff); m[3] = (int8_t)(mm.x >> 24);
        m[4] = (int8_t)(mm.y & 0xff); m[5] = (int8_t)((mm.y >> 8) & 0xff);
        m[6] = (int8_t)((mm.y >> 16) & 0xff); m[7] = (int8_t)(mm.y >> 24);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = min(px0 + i, W - 1);
            m[2 * i] = mrow[(size_t)px * 2u]; m[2 * i + 1] = mrow[(size_t)px * 2u + 1];
        }
    }

    uint32_t o[N][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float uvx = uvxOf[i];
        float mx = (float)m[2 * i], my = (float)m[2 * i + 1];
        if (INTENDED) { mx = mx / (float)W; my = my / (float)H; }
        if (mx == 0.0f && my == 0.0f) {                   // the sample positions do not depend on t
            const V4 p = sample_with_motion(prev, W, H, prevPitch, uvx, uvy, 0.0f, 0.0f, 0.0f);
            const V4 c = sample_with_motion(curr, W, H, currPitch, uvx, uvy, 0.0f, 0.0f, 0.0f);
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float t = tg.t[k];
                o[k][i] = pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
            }
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float t = tg.t[k];
                const V4 p = sample_with_motion(prev, W, H, prevPitch, uvx, uvy, mx, my, -t);
                const V4 c = sample_with_motion(curr, W, H, currPitch, uvx, uvy, mx, my, 1.0f - t);
                o[k][i] = pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        uint8_t *orow = tg.out[k] + (size_t)py * (size_t)tg.pitch[k];
        if (full) {
            *reinterpret_cast<uint4 *>(orow + (size_t)px0 * 4u) = uint4{o[k][0], o[k][1], o[k][2], o[k][3]};
        } else {
            for (int i = 0; i < 4 && px0 + i < W; ++i)
                *reinterpret_cast<uint32_t *>(orow + (size_t)(px0 + i) * 4u) = o[k][i];
        }
    }
}

template <int N>
static void launch_multi_n(hipStream_t s, dim3 grid, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                           const MultiTargets &tg, int W, int H, bool wide, bool intended, const float *uvx, const float *uvy) {
    if (intended)
        hipLaunchKernelGGL((interpolate_multi_kernel<N, true>), grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, tg, W, H, wide, uvx, uvy);
    else
        hipLaunchKernelGGL((interpolate_multi_kernel<N, false>), grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, tg, W, H, wide, uvx, uvy);
}

// `count` frames from one pass per group of up to kMaxMulti factors (a single factor takes interpolate_kernel).
hipError_t launch_interpolate_multi(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                                    const lfg_frame *const *outs, const float *factors, int count, bool intended,
                                    const float *uvx, const float *uvy) {
    const int W = (int)curr.width, H = (int)curr.height;
    const int quads = (W + 3) / 4;
    dim3 grid((quads + 63) / 64, (H + 3) / 4);
    for (int first = 0; first < count; first += kMaxMulti) {
        const int n = count - first < kMaxMulti ? count - first : kMaxMulti;
        if (n == 1) {
            hipError_t e = launch_interpolate(s, prev, curr, mv, *outs[first], factors[first], intended, uvx, uvy);
            if (e != hipSuccess) return e;
            continue;
        }
        MultiTargets tg{};
        bool wide = true;
        for (int k = 0; k < kMaxMulti; ++k) {
            const lfg_frame &o = *outs[first + (k < n ? k : 0)];
            tg.out[k] = (uint8_t *)o.data; tg.pitch[k] = (int)o.pitch; tg.t[k] = factors[first + (k < n ? k : 0)];
            wide = wide && (o.pitch % 16u == 0) && ((uintptr_t)o.data % 16u == 0);
        }
        switch (n) {
            case 2: launch_multi_n<2>(s, grid, prev, curr, mv, tg, W, H, wide, intended, uvx, uvy); break;
            case 3: launch_multi_n<3>(s, grid, prev, curr, mv, tg, W, H, wide, intended, uvx, uvy); break;
            default: launch_multi_n<4>(s, grid, prev, curr, mv, tg, W, H, wide, intended, uvx, uvy); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// vec4(best, 0, 1) per pixel, as shaders/motion.comp:56 stores it.
__global__ __launch_bounds__(256) void mv_export_kernel(const int8_t *__restrict__ mv, int mvPitch,
                                                        float4 *__restrict__ out, int W, int H) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    const int8_t *m = mv + (size_t)py * (size_t)mvPitch + (size_t)px * 2u;
    out[(size_t)py * (size_t)W + (size_t)px] = float4{(float)m[0], (float)m[1], 0.0f, 1.0f};
}

hipError_t launch_mv_export(hipStream_t s, const lfg_frame &mv, float *rgba32f) {
    dim3 grid((mv.width + 63) / 64, (mv.height + 3) / 4);
    hipLaunchKernelGGL(mv_export_kernel, grid, dim3(256), 0, s, (const int8_t *)mv.data, (int)mv.pitch,
                       reinterpret_cast<float4 *>(rgba32f), (int)mv.width, (int)mv.height);
    return hipGetLastError();
}

}  // namespace lfg
