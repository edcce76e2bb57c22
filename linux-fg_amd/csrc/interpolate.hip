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

// ------------------------------------------------------------------------------------------- the quad fast path
//
// A thread's four pixels are four adjacent texels of prev and of curr whenever nothing displaces the samples -- a zero motion
// vector: static content, and under the literal semantics (F5) the ONLY vector whose samples stay inside the image -- and the
// pixel centres' own uv land on texel centres in fp32, which the host has worked out per column and per row (lfg_internal.hpp:
// UvTable::d_centre).  There texture() returns texel (px, py) itself (fraction 0: the other three bilinear products are exactly
// 0), so the thread spends nothing on coordinates, range tests or weights: eight format-converting loads -- an 8_8_8_8 UNORM
// descriptor makes the texture-address unit hand over byte / 255.0f as four floats, bit for bit the oracle's conversion
// (tests/test_gpu_parity.py: test_interpolate_all_byte_values_round_trip) -- then mix and the packing, the same two functions as
// everywhere else.  The columns whose uv misses its texel centre by an ulp (102 of 3840) would make some lane of EVERY wave
// take the bilinear path, so the x axis is walked in the host's order: all-centre quads first, the others together in the
// row's last waves, where the few pixels that need it are recomputed by the generic code.
constexpr int kRsrcRgba8Unorm = (int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (0u << 12) | (10u << 15));

struct QuadTexels { f32x4 p[4], c[4]; };

__device__ __forceinline__ QuadTexels load_quad_texels(const uint8_t *prev, int prevPitch, const uint8_t *curr, int currPitch,
                                                       int H, int px0, int py) {
    const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(prev), 0, H * prevPitch, kRsrcRgba8Unorm);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(curr), 0, H * currPitch, kRsrcRgba8Unorm);
    const int offP = py * prevPitch + px0 * 4, offC = py * currPitch + px0 * 4;
    QuadTexels q;
#pragma unroll
    for (int k = 0; k < 4; ++k) q.p[k] = buffer_load_rgba8_format(rP, offP + 4 * k, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) q.c[k] = buffer_load_rgba8_format(rC, offC + 4 * k, 0);
    return q;
}

__device__ __forceinline__ uint32_t mix_pack(const f32x4 p, const f32x4 c, float t) {
    return pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
}

// Which quad this thread works on, and whether its wave holds all-centre quads only.
__device__ __forceinline__ int quad_of_slot(const InterpTables &tb, int slot) {
    return tb.quads ? (int)tb.quads[slot] : slot;
}

// The host's verdict on whether the fast path may be used for these frames at all (launch_interpolate*): every row pitch and
// base address aligned for the 8-byte vector loads and the 16-byte stores, the frames below 2 GiB (32-bit buffer offsets).
static bool fast_frames(const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv) {
    const auto small = [](const lfg_frame &f) { return (uint64_t)f.pitch * f.height < (1ull << 31); };
    return small(prev) && small(curr) && (uintptr_t)prev.data % 4u == 0 && (uintptr_t)curr.data % 4u == 0 &&
           mv.pitch % 8u == 0 && (uintptr_t)mv.data % 8u == 0;
}
static bool fast_output(const lfg_frame &o) { return o.pitch % 16u == 0 && (uintptr_t)o.data % 16u == 0; }

// INTENDED = false: interpolate.comp as written.  INTENDED = true (opt-in, lfg_set_semantics):
// the motion vector is divided by the image size before it is added to uv, so it displaces by pixels.
template <bool INTENDED>
__global__ __launch_bounds__(256) void interpolate_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    const int8_t *__restrict__ mv, int mvPitch, uint8_t *__restrict__ out, int outPitch,
    int W, int H, float t, InterpTables tb, int fast) {
    const int slot = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (py >= H) return;
    const int qx = quad_of_slot(tb, slot);                       // group of four pixels
    const int px0 = qx * 4;
    if (px0 >= W) return;
    const int8_t *mrow = mv + (size_t)py * (size_t)mvPitch;
    uint8_t *orow = out + (size_t)py * (size_t)outPitch;
    const bool full = (px0 + 3 < W) && ((mvPitch & 7) == 0) && ((outPitch & 15) == 0);

    int8_t m[8];
    uint2 mm = uint2{1u, 1u};
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
        mm = *reinterpret_cast<const uint2 *>(mrow + (size_t)px0 * 2u);
        if (fast && (mm.x | mm.y) == 0u && tb.centreY[py]) {
            // four zero vectors in a row whose own uv is a texel centre: the quad fast path (see above)
            const QuadTexels q = load_quad_texels(prev, prevPitch, curr, currPitch, H, px0, py);
            uint32_t centres = 0x01010101u;
            if ((int)blockIdx.x * 64 >= tb.goodSlots) centres = *reinterpret_cast<const uint32_t *>(tb.centreX + px0);
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = mix_pack(q.p[i], q.c[i], t);
            if (centres != 0x01010101u) {                         // (the row's last waves only)
                const float uvy = tb.uvy[py];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (((centres >> (8 * i)) & 1u) == 0u) {
                        const float uvx = tb.uvx[px0 + i];
                        const V4 p = sample_with_motion(prev, W, H, prevPitch, uvx, uvy, 0.0f, 0.0f, -t);
                        const V4 c = sample_with_motion(curr, W, H, currPitch, uvx, uvy, 0.0f, 0.0f, 1.0f - t);
                        o[i] = pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
                    }
            }
            *reinterpret_cast<uint4 *>(orow + (size_t)px0 * 4u) = uint4{o[0], o[1], o[2], o[3]};
            return;
        }
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

    // ((float)p + 0.5f) / (float)size, interpolate.comp:30, from the host's tables (lfg_internal.hpp: UvTable; the tables are
    // padded to whole groups of four)
    const float uvy = tb.uvy[py];
    const float4 uvx4 = *reinterpret_cast<const float4 *>(tb.uvx + px0);
    const float uvxOf[4] = {uvx4.x, uvx4.y, uvx4.z, uvx4.w};
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

static dim3 interpolate_grid(const InterpTables &tb, int W, int H) {
    const int quads = (W + 3) / 4;
    return dim3(tb.quads ? tb.quadSlots / 64 : (quads + 63) / 64, (H + 3) / 4);
}

hipError_t launch_interpolate(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                              const lfg_frame &mv, const lfg_frame &out, float factor, bool intended, const InterpTables &tb) {
    const dim3 grid = interpolate_grid(tb, (int)out.width, (int)out.height);
    const int fast = fast_frames(prev, curr, mv) && fast_output(out) ? 1 : 0;
    if (intended)
        hipLaunchKernelGGL(interpolate_kernel<true>, grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, (uint8_t *)out.data, (int)out.pitch,
                           (int)out.width, (int)out.height, factor, tb, fast);
    else
        hipLaunchKernelGGL(interpolate_kernel<false>, grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, (uint8_t *)out.data, (int)out.pitch,
                           (int)out.width, (int)out.height, factor, tb, fast);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ several factors per pair
//
// BASELINE config 5 (60 -> 240 fps) generates three frames per pair, t = 1/4, 1/2, 3/4, from ONE motion field.  Run
// as three dispatches of the kernel above, prev, curr and the vectors are read three times: 3 x 14 = 42 B/pixel.
// Here one thread reads its four vectors once and produces the pixels of all N frames: 10 + 4N B/pixel (22 for N = 3).
// Where the vector is (0,0) -- static content, and under the literal semantics the only case in which both sources
// are sampled inside the image for every t -- the two texels are fetched once and only mix() is repeated (the quad fast
// path above where it applies).
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
    InterpTables tb, int fast) {
    const int slot = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (py >= H) return;
    const int qx = quad_of_slot(tb, slot);
    const int px0 = qx * 4;
    if (px0 >= W) return;
    const int8_t *mrow = mv + (size_t)py * (size_t)mvPitch;
    const bool full = (px0 + 3 < W) && ((mvPitch & 7) == 0) && wideStores;

    int8_t m[8];
    if (full) {
        const uint2 mm = *reinterpret_cast<const uint2 *>(mrow + (size_t)px0 * 2u);
        if (fast && (mm.x | mm.y) == 0u && tb.centreY[py]) {
            const QuadTexels q = load_quad_texels(prev, prevPitch, curr, currPitch, H, px0, py);
            uint32_t centres = 0x01010101u;
            if ((int)blockIdx.x * 64 >= tb.goodSlots) centres = *reinterpret_cast<const uint32_t *>(tb.centreX + px0);
            uint32_t o[N][4];
#pragma unroll
            for (int k = 0; k < N; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) o[k][i] = mix_pack(q.p[i], q.c[i], tg.t[k]);
            if (centres != 0x01010101u) {
                const float uvy = tb.uvy[py];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (((centres >> (8 * i)) & 1u) == 0u) {
                        const float uvx = tb.uvx[px0 + i];
                        const V4 p = sample_with_motion(prev, W, H, prevPitch, uvx, uvy, 0.0f, 0.0f, 0.0f);
                        const V4 c = sample_with_motion(curr, W, H, currPitch, uvx, uvy, 0.0f, 0.0f, 0.0f);
#pragma unroll
                        for (int k = 0; k < N; ++k) {
                            const float t = tg.t[k];
                            o[k][i] = pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
                        }
                    }
            }
#pragma unroll
            for (int k = 0; k < N; ++k)
                *reinterpret_cast<uint4 *>(tg.out[k] + (size_t)py * (size_t)tg.pitch[k] + (size_t)px0 * 4u) = uint4{o[k][0], o[k][1], o[k][2], o[k][3]};
            return;
        }
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

    const float uvy = tb.uvy[py];
    const float4 uvx4 = *reinterpret_cast<const float4 *>(tb.uvx + px0);
    const float uvxOf[4] = {uvx4.x, uvx4.y, uvx4.z, uvx4.w};
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
                           const MultiTargets &tg, int W, int H, bool wide, bool intended, const InterpTables &tb, int fast) {
    if (intended)
        hipLaunchKernelGGL((interpolate_multi_kernel<N, true>), grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, tg, W, H, wide, tb, fast);
    else
        hipLaunchKernelGGL((interpolate_multi_kernel<N, false>), grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, tg, W, H, wide, tb, fast);
}

// `count` frames from one pass per group of up to kMaxMulti factors (a single factor takes interpolate_kernel).
hipError_t launch_interpolate_multi(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                                    const lfg_frame *const *outs, const float *factors, int count, bool intended,
                                    const InterpTables &tb) {
    const int W = (int)curr.width, H = (int)curr.height;
    const dim3 grid = interpolate_grid(tb, W, H);
    for (int first = 0; first < count; first += kMaxMulti) {
        const int n = count - first < kMaxMulti ? count - first : kMaxMulti;
        if (n == 1) {
            hipError_t e = launch_interpolate(s, prev, curr, mv, *outs[first], factors[first], intended, tb);
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
        const int fast = fast_frames(prev, curr, mv) && wide ? 1 : 0;
        switch (n) {
            case 2: launch_multi_n<2>(s, grid, prev, curr, mv, tg, W, H, wide, intended, tb, fast); break;
            case 3: launch_multi_n<3>(s, grid, prev, curr, mv, tg, W, H, wide, intended, tb, fast); break;
            default: launch_multi_n<4>(s, grid, prev, curr, mv, tg, W, H, wide, intended, tb, fast); break;
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
