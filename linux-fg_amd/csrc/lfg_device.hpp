// Device-side helpers shared by the three kernels.  gfx950 only.
//
// The whole library is compiled with -ffp-contract=off: an fp32 multiply followed by an add stays
// two roundings unless the source says __builtin_fmaf.  The motion and interpolate kernels rely on
// that to reproduce the reference shaders' arithmetic (as fixed in SURVEY.md section 8(c)) bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace lfg {

// 1/255 split in two floats: fma(k, kInv255Hi, k * kInv255Lo) == (float)k / 255.0f for every
// integer k in [0,255] (checked exhaustively on the host in tests/test_host_logic.py and on the
// device by tests/test_gpu_parity.py::test_unorm_conversion_exact).  Two VALU ops instead of a
// correctly rounded division.
__device__ constexpr float kInv255Hi = 0x1.010102p-8f;     // RN(1/255) = 0.003921568859368563
__device__ constexpr float kInv255Lo = -0x1.fdfdfep-33f;    // RN(1/255 - kInv255Hi) = -2.319175823606301e-10

// UNORM8 -> float, identical to (float)b / 255.0f.
__device__ __forceinline__ float unorm8_to_float(float k) {
    return __builtin_fmaf(k, kInv255Hi, k * kInv255Lo);
}

__device__ __forceinline__ float byte0(uint32_t p) { return (float)(p & 0xffu); }
__device__ __forceinline__ float byte1(uint32_t p) { return (float)((p >> 8) & 0xffu); }
__device__ __forceinline__ float byte2(uint32_t p) { return (float)((p >> 16) & 0xffu); }
__device__ __forceinline__ float byte3(uint32_t p) { return (float)(p >> 24); }

// A value already in 0..255 scale -> byte: clamp, round half to even (v_rndne_f32).
__device__ __forceinline__ uint32_t to_byte_255(float v) {
    v = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 255.0f);       // fmax(NaN, 0) = 0
    return (uint32_t)__builtin_rintf(v);
}

__device__ __forceinline__ uint32_t pack_rgba8_255(float r, float g, float b, float a) {
    return to_byte_255(r) | (to_byte_255(g) << 8) | (to_byte_255(b) << 16) | (to_byte_255(a) << 24);
}

// A value in 0..1 scale -> UNORM8 exactly as the oracle stores it: clamp to [0,1], * 255, RNE.
__device__ __forceinline__ uint32_t to_unorm8(float v) {
    v = __builtin_fminf(__builtin_fmaxf(v, 0.0f), 1.0f);
    return (uint32_t)__builtin_rintf(v * 255.0f);
}

// Orders this wave's LDS traffic for cross-lane exchange inside ONE wave: a wavefront-scope fence
// keeps the compiler from moving a lane's LDS reads above its own LDS write (they never alias for
// the same lane, but they do across lanes); the hardware executes one wave's LDS ops in order.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

}  // namespace lfg
