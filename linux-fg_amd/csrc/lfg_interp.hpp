// The per-pixel arithmetic of shaders/interpolate.comp:15-40, shared by interpolate.hip (the stage itself) and
// scale.hip (the fused interpolate -> 2x scale of the input-resolution data flow): one definition, so both produce
// the same bytes.  Operation order as fixed in SURVEY.md section 8(c); the library is built with -ffp-contract=off.
#pragma once

#include "lfg_device.hpp"

namespace lfg {

struct V4 { float x, y, z, w; };

__device__ __forceinline__ V4 texel_unorm(const uint8_t *__restrict__ img, int pitch, int x, int y) {
    const uint32_t p = *reinterpret_cast<const uint32_t *>(img + (size_t)y * (size_t)pitch + (size_t)x * 4u);
    return V4{unorm8_to_float(byte0(p)), unorm8_to_float(byte1(p)),
              unorm8_to_float(byte2(p)), unorm8_to_float(byte3(p))};
}

// texture() on a LINEAR / CLAMP_TO_EDGE sampler with normalised coordinates
// (src/frame_manager.cpp:555-561), oracle choice (3).
__device__ __forceinline__ V4 texture_bilinear(const uint8_t *__restrict__ img, int W, int H, int pitch,
                                               float s, float t) {
    const float u = s * (float)W - 0.5f;
    const float v = t * (float)H - 0.5f;
    const float fu = __builtin_floorf(u), fv = __builtin_floorf(v);
    const float a = u - fu, b = v - fv;
    int i0 = (int)fu, j0 = (int)fv;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = clampi(i0, 0, W - 1); i1 = clampi(i1, 0, W - 1);
    j0 = clampi(j0, 0, H - 1); j1 = clampi(j1, 0, H - 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b);
    const float w01 = (1.0f - a) * b, w11 = a * b;
    const V4 t00 = texel_unorm(img, pitch, i0, j0);
    if (a == 0.0f && b == 0.0f) return t00;     // w00 == 1, the other three products are exactly 0
    const V4 t10 = texel_unorm(img, pitch, i1, j0);
    const V4 t01 = texel_unorm(img, pitch, i0, j1);
    const V4 t11 = texel_unorm(img, pitch, i1, j1);
    V4 r;
    r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
    r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
    r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
    r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
    return r;
}

// interpolate.comp:15-22
__device__ __forceinline__ V4 sample_with_motion(const uint8_t *__restrict__ img, int W, int H, int pitch,
                                                 float uvx, float uvy, float mx, float my, float scale) {
    const float sx = uvx + mx * scale, sy = uvy + my * scale;
    if (sx < 0.0f || sy < 0.0f || sx > 1.0f || sy > 1.0f) return V4{0.f, 0.f, 0.f, 0.f};
    return texture_bilinear(img, W, H, pitch, sx, sy);
}

__device__ __forceinline__ float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }

// One interpolated pixel as packed RGBA8: mix(S(prev, uv - mv t), S(curr, uv + mv (1 - t)), t), clamped, x 255,
// round half to even (interpolate.comp:30-39).  mx, my: the motion vector in pixels (already divided by the image
// size under the opt-in intended semantics).
__device__ __forceinline__ uint32_t interpolate_pixel(const uint8_t *__restrict__ prev, int prevPitch,
                                                      const uint8_t *__restrict__ curr, int currPitch,
                                                      int W, int H, int px, int py, float mx, float my, float t) {
    const float uvx = ((float)px + 0.5f) / (float)W, uvy = ((float)py + 0.5f) / (float)H;
    const V4 p = sample_with_motion(prev, W, H, prevPitch, uvx, uvy, mx, my, -t);
    const V4 c = sample_with_motion(curr, W, H, currPitch, uvx, uvy, mx, my, 1.0f - t);
    return pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
}

}  // namespace lfg
