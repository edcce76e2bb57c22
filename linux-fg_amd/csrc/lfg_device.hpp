// Device-side helpers shared by the three kernels.  gfx950 only.
//
// The whole library is compiled with -ffp-contract=off: an fp32 multiply followed by an add stays
// two roundings unless the source says __builtin_fmaf.  The motion and interpolate kernels rely on
// that to reproduce the reference shaders' arithmetic (as fixed in SURVEY.md section 8(c)) bit for bit.
#pragma once
#include <type_traits>

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

// v_cvt_pk_u8_f32 converts with round-half-to-even and saturates to [0,255] (NaN -> 0); measured on
// gfx950 by tools/microbench.hip: 0.5->0, 1.5->2, 2.5->2, 254.5->254, 300->255, -3->0.  One VALU op
// per channel replaces clamp + rndne + cvt + shift/or.
__device__ __forceinline__ uint32_t pack_rgba8_255(float r, float g, float b, float a) {
    uint32_t p = __builtin_amdgcn_cvt_pk_u8_f32(r, 0u, 0u);
    p = __builtin_amdgcn_cvt_pk_u8_f32(g, 1u, p);
    p = __builtin_amdgcn_cvt_pk_u8_f32(b, 2u, p);
    return __builtin_amdgcn_cvt_pk_u8_f32(a, 3u, p);
}

// Values in 0..1 scale -> packed UNORM8 exactly as the oracle stores them: clamp to [0,1], * 255,
// round half to even.  Scaling first and saturating in the conversion gives the same byte.
__device__ __forceinline__ uint32_t pack_rgba8_unorm(float r, float g, float b, float a) {
    return pack_rgba8_255(r * 255.0f, g * 255.0f, b * 255.0f, a * 255.0f);
}

// ---- typed buffer loads: the texture-address unit converts RGBA8 UNORM to four floats on the way in.
// Measured on gfx950 (tools/probe_unorm.hip): the conversion is bit-exact (float)k / 255.0f for all
// 256 byte values, i.e. oracle choice (1), at zero VALU cost.  hipcc exposes no builtin for the format
// loads, so they are issued from inline asm; the compiler does not track them, hence the explicit
// s_waitcnt that takes every loaded register as an in/out operand (uses cannot be scheduled above it).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Buffer resource (V#) over `bytes` bytes at `base`: raw (stride 0), DST_SEL = R,G,B,A,
// NUM_FORMAT = UNORM (0), DATA_FORMAT = 8_8_8_8 (10).  Offsets >= bytes (so also "negative" offsets,
// which wrap to >= 2^31) are out of range and load (0,0,0,0): oracle choice (5) for rows above/below
// the image.  The four dwords are made wave-uniform for the SGPR operand.
__device__ __forceinline__ i32x4 make_rgba8_rsrc(const void *base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(b >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = __builtin_amdgcn_readfirstlane((int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (0u << 12) | (10u << 15)));
    return r;
}

// Load ten RGBA8 texels at byte offsets `o[]` as float4 each.  Issue and wait live in ONE asm statement:
// the outputs are only defined once the statement ends, so the compiler can never copy or spill a
// register that a load is still writing (it does not track these loads).
__device__ __forceinline__ void load_rgba8_unorm_x10(f32x4 (&p)[10], const int (&o)[10], i32x4 rsrc) {
    asm volatile(
        "tbuffer_load_format_xyzw %0, %10, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %1, %11, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %2, %12, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %3, %13, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %4, %14, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %5, %15, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %6, %16, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %7, %17, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %8, %18, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "tbuffer_load_format_xyzw %9, %19, %20, 0 format:[BUF_DATA_FORMAT_8_8_8_8,BUF_NUM_FORMAT_UNORM] offen\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]),
          "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(p[8]), "=&v"(p[9])
        : "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(o[4]), "v"(o[5]), "v"(o[6]), "v"(o[7]), "v"(o[8]), "v"(o[9]),
          "s"(rsrc)
        : "memory");
}

// ---- format-converting buffer loads the compiler DOES track (vmcnt bookkeeping, scheduling): the LLVM intrinsic
// llvm.amdgcn.raw.ptr.buffer.load.format.v4f32, reached by declaring a function under its name -- hipcc has no
// __builtin for the format loads.  The conversion comes from the descriptor (word 3): DST_SEL = R,G,B,A,
// DATA_FORMAT = 8_8_8_8 (10), NUM_FORMAT = USCALED (2): each byte k arrives as the float (float)k, exactly
// (tests/test_gpu_parity.py::test_format_load_uscaled_exact), so a kernel that wants its texels on a 0..255 scale
// spends no VALU instruction on unpacking.
constexpr int kRsrcRaw32 = 0x00020000;                                     // DATA_FORMAT = 32: plain raw buffer
constexpr int kRsrcRgba8Uscaled = (int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (2u << 12) | (10u << 15));
extern "C" __device__ f32x4 lfg_llvm_raw_ptr_buffer_load_format_v4f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.format.v4f32");
// One RGBA8 texel at byte offset voffset (per lane, range-checked: out of range loads 0) + soffset (wave-uniform,
// NOT range-checked) as four floats in 0..255.
__device__ __forceinline__ f32x4 buffer_load_rgba8_format(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
    return lfg_llvm_raw_ptr_buffer_load_format_v4f32(rsrc, voffset, soffset, 0);
}

// 16-byte buffer store + the wait states a store of more than 8 bytes needs before its data registers may be
// rewritten.  The ISA manual asks for one wait state and exempts stores with a scalar offset; gfx950 showed the
// exempted form losing that race (the following VALU write reached a few pixels per 4K frame, not every run).
// Here the row offset travels in the vector offset (soffset = 0), so the compiler's hazard recogniser covers the
// store as documented, two more wait states follow in any case, and csrc/Makefile's check_store_hazard.py fails the
// build if any dwordx4 store in the object is not followed by them.
typedef unsigned int u32x4_store __attribute__((ext_vector_type(4)));
template <int AUX = 2>
__device__ __forceinline__ void store_b128_guarded(u32x4_store q, __amdgpu_buffer_rsrc_t rsrc, int voffset) {
    __builtin_amdgcn_raw_buffer_store_b128(q, rsrc, voffset, 0, AUX);
    // The data registers are an INPUT of the wait states: the compiler must keep them intact until the s_nop has
    // issued, and it does not move a volatile asm across the store.
#ifndef LFG_DIAG_NO_STORE_GUARD      // (diagnostic build of tools/repro_store_hazard.py only: the store without its wait states)
    asm volatile("s_nop 1" : : "v"(q));
#endif
}

// Orders this wave's LDS traffic for cross-lane exchange inside ONE wave: a wavefront-scope fence
// keeps the compiler from moving a lane's LDS reads above its own LDS write (they never alias for
// the same lane, but they do across lanes); the hardware executes one wave's LDS ops in order.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// The largest of a wave's 64 unsigned values, as a wave-uniform (scalar) result.  Six DPP steps in the VALU -- within
// quads (quad_perm), within half rows and rows (row_half_mirror, row_mirror), then across rows (row_bcast15, row_bcast31:
// lane 63 ends up with everything) -- instead of six ds_bpermute round trips through the LDS pipeline, which is what
// __shfl_xor compiles to (6 x ~120 cycles of latency in a chain).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    // __builtin_amdgcn_update_dpp(old, src, dpp_ctrl, row_mask, bank_mask, bound_ctrl): lanes whose source is invalid keep
    // `old` = 0, the identity of max over unsigned values
    auto step = [](uint32_t x, auto ctrl, auto rowMask) {
        const uint32_t moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, decltype(rowMask)::value, 0xF, false);
        return x > moved ? x : moved;
    };
    v = step(v, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{});    // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{});    // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{});   // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{});   // row_mirror
    v = step(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});   // row_bcast15 into rows 1 and 3
    v = step(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});   // row_bcast31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

}  // namespace lfg
