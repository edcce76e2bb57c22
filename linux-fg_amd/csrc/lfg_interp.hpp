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

// ---- the same sample, in two steps, for code that wants all of a thread's loads in flight before it uses any of them
// (interpolate.hip's generic path: round 3's per-sample branches -- outside [0,1]?  fractions zero? -- put up to sixteen memory
// round trips in a row into one wave, and the few waves that needed them set the kernel's duration).  No shortcut for zero
// fractions: with a = b = 0 the weights are 1, 0, 0, 0 and the sum below is t00 + 0 + 0 + 0, the same float.
struct SampleTaps {
    const uint8_t *r0, *r1;      // the two texel rows (clamped), at column 0
    int i0, i1;                  // the two texel columns (clamped)
    float w00, w10, w01, w11;
    bool inside;                 // interpolate.comp:17-20: outside [0,1] the sample is vec4(0)
};

__device__ __forceinline__ SampleTaps sample_taps(const uint8_t *__restrict__ img, int W, int H, int pitch,
                                                  float uvx, float uvy, float mx, float my, float scale) {
    const float sx = uvx + mx * scale, sy = uvy + my * scale;
    SampleTaps s;
    s.inside = !(sx < 0.0f || sy < 0.0f || sx > 1.0f || sy > 1.0f);
    const float u = sx * (float)W - 0.5f;
    const float v = sy * (float)H - 0.5f;
    const float fu = __builtin_floorf(u), fv = __builtin_floorf(v);
    const float a = u - fu, b = v - fv;
    // (a sample far outside the image may not fit an int: it is not used, only its addresses have to be valid)
    const int i0 = (int)__builtin_fminf(__builtin_fmaxf(fu, -2.0f), (float)W);
    const int j0 = (int)__builtin_fminf(__builtin_fmaxf(fv, -2.0f), (float)H);
    s.i0 = clampi(i0, 0, W - 1); s.i1 = clampi(i0 + 1, 0, W - 1);
    s.r0 = img + (size_t)clampi(j0, 0, H - 1) * (size_t)pitch;
    s.r1 = img + (size_t)clampi(j0 + 1, 0, H - 1) * (size_t)pitch;
    s.w00 = (1.0f - a) * (1.0f - b); s.w10 = a * (1.0f - b);
    s.w01 = (1.0f - a) * b; s.w11 = a * b;
    return s;
}

// interpolate.comp:16-20 alone: does the displaced sample lie inside [0,1] x [0,1]?  (The same two sums as sample_taps.)
__device__ __forceinline__ bool sample_inside(float uvx, float uvy, float mx, float my, float scale) {
    const float sx = uvx + mx * scale, sy = uvy + my * scale;
    return !(sx < 0.0f || sy < 0.0f || sx > 1.0f || sy > 1.0f);
}

struct SampleTexels { uint32_t t00, t10, t01, t11; };

__device__ __forceinline__ SampleTexels sample_load(const SampleTaps &s) {
    SampleTexels t;
    t.t00 = *reinterpret_cast<const uint32_t *>(s.r0 + (size_t)s.i0 * 4u);
    t.t10 = *reinterpret_cast<const uint32_t *>(s.r0 + (size_t)s.i1 * 4u);
    t.t01 = *reinterpret_cast<const uint32_t *>(s.r1 + (size_t)s.i0 * 4u);
    t.t11 = *reinterpret_cast<const uint32_t *>(s.r1 + (size_t)s.i1 * 4u);
    return t;
}

__device__ __forceinline__ V4 unorm4(uint32_t p) {
    return V4{unorm8_to_float(byte0(p)), unorm8_to_float(byte1(p)), unorm8_to_float(byte2(p)), unorm8_to_float(byte3(p))};
}

__device__ __forceinline__ V4 sample_finish(const SampleTaps &s, const SampleTexels &t) {
    const V4 t00 = unorm4(t.t00), t10 = unorm4(t.t10), t01 = unorm4(t.t01), t11 = unorm4(t.t11);
    V4 r;
    r.x = ((s.w00 * t00.x + s.w10 * t10.x) + s.w01 * t01.x) + s.w11 * t11.x;
    r.y = ((s.w00 * t00.y + s.w10 * t10.y) + s.w01 * t01.y) + s.w11 * t11.y;
    r.z = ((s.w00 * t00.z + s.w10 * t10.z) + s.w01 * t01.z) + s.w11 * t11.z;
    r.w = ((s.w00 * t00.w + s.w10 * t10.w) + s.w01 * t01.w) + s.w11 * t11.w;
    if (!s.inside) r = V4{0.f, 0.f, 0.f, 0.f};
    return r;
}

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
