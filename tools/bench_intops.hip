// Issue-rate microbenchmark for the integer / transcendental VALU ops the motion prefilter is built from
// (gfx950): 16 independent accumulators per lane, 2 waves per SIMD.  Prints lane-ops/s relative to v_add_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define CHAIN_KERNEL(NAME, ASMSTR)                                                                     \
    template <int ITERS> __global__ void NAME(unsigned *out, unsigned seed) {                          \
        unsigned a[16];                                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = seed * (unsigned)(i + 1 + threadIdx.x);  \
        unsigned x = seed + threadIdx.x, y = seed * 3u + 1u;                                           \
        for (int it = 0; it < ITERS; ++it) {                                                           \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                            \
                _Pragma("unroll") for (int i = 0; i < 16; ++i)                                         \
                    asm volatile(ASMSTR : "+v"(a[i]) : "v"(x), "v"(y) : "s40", "s41", "s42", "s43", "vcc", "scc");                                \
            }                                                                                          \
        }                                                                                              \
        unsigned s = 0;                                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i];                                      \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
    }

CHAIN_KERNEL(k_add_f32, "v_add_f32 %0, %0, %1")
CHAIN_KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
CHAIN_KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
CHAIN_KERNEL(k_sad, "v_sad_u8 %0, %1, %2, %0")
CHAIN_KERNEL(k_mad24, "v_mad_i32_i24 %0, %1, -2, %0")
CHAIN_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %1, 1, %0")
CHAIN_KERNEL(k_sqrt, "v_sqrt_f32 %0, %0")
CHAIN_KERNEL(k_rsq, "v_rsq_f32 %0, %0")
CHAIN_KERNEL(k_cvt, "v_cvt_f32_u32 %0, %0")
CHAIN_KERNEL(k_min, "v_min_f32 %0, %0, %1")
CHAIN_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
CHAIN_KERNEL(k_sqrt_add, "v_sqrt_f32 %0, %0\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %2\n v_add_f32 %0, %0, %1")
CHAIN_KERNEL(k_mul_f32, "v_mul_f32 %0, %0, %1")
CHAIN_KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
CHAIN_KERNEL(k_sub_f32, "v_sub_f32 %0, %0, %1")
CHAIN_KERNEL(k_sub_u32, "v_sub_u32 %0, %0, %1")
CHAIN_KERNEL(k_lshlrev, "v_lshlrev_b32 %0, 1, %0")
CHAIN_KERNEL(k_and, "v_and_b32 %0, %0, %1")
CHAIN_KERNEL(k_max_f32, "v_max_f32 %0, %0, %1")
CHAIN_KERNEL(k_mov, "v_mov_b32 %0, %1")
CHAIN_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
CHAIN_KERNEL(k_cvt_i32, "v_cvt_f32_i32 %0, %0")
CHAIN_KERNEL(k_cvt_ub0, "v_cvt_f32_ubyte0 %0, %0")
CHAIN_KERNEL(k_cmp_vcc, "v_cmp_le_f32 vcc, %0, %1")
CHAIN_KERNEL(k_cmp_sgpr, "v_cmp_le_f32 s[40:41], %0, %1")
CHAIN_KERNEL(k_cnd_sgpr, "v_cndmask_b32 %0, %0, %1, s[40:41]")
CHAIN_KERNEL(k_cmp_cnd, "v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
CHAIN_KERNEL(k_min3, "v_min3_f32 %0, %0, %1, %2")
CHAIN_KERNEL(k_add_salu, "v_add_f32 %0, %0, %1\n s_add_u32 s40, s40, 1")
CHAIN_KERNEL(k_add_2salu, "v_add_f32 %0, %0, %1\n s_add_u32 s40, s40, 1\n s_and_b64 s[42:43], s[42:43], exec")
CHAIN_KERNEL(k_dot_salu, "v_dot4_u32_u8 %0, %1, %2, %0\n s_add_u32 s40, s40, 1")
CHAIN_KERNEL(k_add_nop, "v_add_f32 %0, %0, %1\n s_nop 0")
CHAIN_KERNEL(k_add_branch, "v_add_f32 %0, %0, %1\n s_cbranch_execz 1f\n 1:")
CHAIN_KERNEL(k_cmp_saveexec, "v_cmp_le_f32 vcc, %0, %1\n s_and_saveexec_b64 s[42:43], vcc\n s_cbranch_execz 2f\n 2: s_or_b64 exec, exec, s[42:43]")
// round 2: candidates for the scale kernel's filter arithmetic and packing
CHAIN_KERNEL(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %1, %2, %0")
CHAIN_KERNEL(k_dot2_i32_i16, "v_dot2_i32_i16 %0, %1, %2, %0")
CHAIN_KERNEL(k_dot4_i32_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
CHAIN_KERNEL(k_dot2c_f32_f16, "v_dot2c_f32_f16 %0, %1, %2")
CHAIN_KERNEL(k_pk_fma_f16, "v_pk_fma_f16 %0, %1, %2, %0")
CHAIN_KERNEL(k_pk_mul_f16, "v_pk_mul_f16 %0, %0, %1")
CHAIN_KERNEL(k_pk_mad_i16, "v_pk_mad_i16 %0, %1, %2, %0")
CHAIN_KERNEL(k_fmac_f32, "v_fmac_f32 %0, %1, %2")
CHAIN_KERNEL(k_cvt_pk_u8, "v_cvt_pk_u8_f32 %0, %1, 1, %0")
CHAIN_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
CHAIN_KERNEL(k_med3, "v_med3_f32 %0, %0, %1, %2")
CHAIN_KERNEL(k_cvt_ub1, "v_cvt_f32_ubyte1 %0, %0")
CHAIN_KERNEL(k_cvt_pkrtz, "v_cvt_pkrtz_f16_f32 %0, %0, %1")
CHAIN_KERNEL(k_fma_mix, "v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]")
CHAIN_KERNEL(k_mad_u32_u16, "v_mad_u32_u16 %0, %1, %2, %0")
CHAIN_KERNEL(k_fma_dpp, "v_fmac_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
CHAIN_KERNEL(k_dot4_add, "v_dot4_u32_u8 %0, %1, %2, %0\n v_add_u32 %0, %0, %1")

template <typename K>
static double time_kernel(K launch, int reps) {
    hipEvent_t b, e;
    hipEventCreate(&b); hipEventCreate(&e);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(b);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms = 0;
    hipEventElapsedTime(&ms, b, e);
    return ms / reps;
}

int main() {
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned *out;
    CK(hipMalloc(&out, 256 * 16 * 1024 * sizeof(unsigned)));
    constexpr int IT = 1000;
    for (int wps : {2, 3}) {
        const int blocks = prop.multiProcessorCount * wps;
        const double insts = (double)blocks * 256 * IT * 8 * 16;
#define RUN(K, PER) { double t = time_kernel([&] { hipLaunchKernelGGL(K<IT>, dim3(blocks), dim3(256), 0, 0, out, 3u); }, 5); \
                      printf("%-12s waves/SIMD %d: %7.3f ms  %6.2f T lane-insts/s\n", #K, wps, t, insts * PER / t / 1e9); }
        RUN(k_add_f32, 1) RUN(k_add_u32, 1) RUN(k_dot4, 1) RUN(k_sad, 1) RUN(k_mad24, 1) RUN(k_lshl_add, 1)
        RUN(k_sqrt, 1) RUN(k_rsq, 1) RUN(k_cvt, 1) RUN(k_min, 1) RUN(k_cndmask, 1) RUN(k_sqrt_add, 4) RUN(k_dot4_add, 2)
        RUN(k_mul_f32, 1) RUN(k_fma_f32, 1) RUN(k_sub_f32, 1) RUN(k_sub_u32, 1) RUN(k_lshlrev, 1) RUN(k_and, 1) RUN(k_max_f32, 1)
        RUN(k_mov, 1) RUN(k_add3, 1) RUN(k_cvt_i32, 1) RUN(k_cvt_ub0, 1) RUN(k_cmp_vcc, 1) RUN(k_cmp_sgpr, 1) RUN(k_cnd_sgpr, 1)
        RUN(k_cmp_cnd, 2) RUN(k_min3, 1)
        RUN(k_dot2_f32_f16, 1) RUN(k_dot2_i32_i16, 1) RUN(k_dot4_i32_i8, 1) RUN(k_dot2c_f32_f16, 1) RUN(k_pk_fma_f16, 1) RUN(k_pk_mul_f16, 1)
        RUN(k_pk_mad_i16, 1) RUN(k_fmac_f32, 1) RUN(k_cvt_pk_u8, 1) RUN(k_perm, 1) RUN(k_med3, 1) RUN(k_cvt_ub1, 1) RUN(k_cvt_pkrtz, 1)
        RUN(k_fma_mix, 1) RUN(k_mad_u32_u16, 1) RUN(k_fma_dpp, 1)
        RUN(k_add_salu, 1) RUN(k_add_2salu, 1) RUN(k_dot_salu, 1) RUN(k_add_nop, 1) RUN(k_add_branch, 1) RUN(k_cmp_saveexec, 1)
    }
    return 0;
}
