// Issue cost of the instructions the strip kernel (csrc/motion_strip.hip) is made of, one wave per SIMD and four:
// cycles per instruction of 16 independent chains.   hipcc -O3 --offload-arch=gfx950 tools/bench_dpp.hip -o tools/bench_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ float shl1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true)); }
__device__ __forceinline__ float rowshl1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x101, 0xF, 0xF, true)); }
template <int OP>
__global__ void k(float *out, unsigned long long *cyc, int iters) {
    float x[16]; unsigned u[16];
    for (int i = 0; i < 16; ++i) { x[i] = threadIdx.x * 0.5f + i; u[i] = threadIdx.x * 7u + i; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (OP == 0) x[i] = x[i] + 1.25f;
            if (OP == 1) x[i] = x[i] + shl1(x[i]);                      // v_add_f32_dpp wave_shl:1
            if (OP == 2) x[i] = shl1(x[i]);                             // v_mov_b32_dpp wave_shl:1
            if (OP == 3) x[i] = x[i] + rowshl1(x[i]);                   // v_add_f32_dpp row_shl:1
            if (OP == 4) x[i] = __builtin_amdgcn_sqrtf(x[i]);
            if (OP == 5) u[i] = __builtin_amdgcn_udot4(u[i], u[i], u[i], false);
            if (OP == 6) { unsigned d; asm volatile("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(u[i]), "v"(u[(i + 1) & 15])); u[i] = d; }
            if (OP == 7) u[i] = min(u[i], u[(i + 3) & 15] + 1u);
            if (OP == 8) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
            if (OP == 9) { typedef float f2 __attribute__((ext_vector_type(2))); if (i < 8) { f2 v = {x[2 * i], x[2 * i + 1]}; v = __builtin_elementwise_fma(v, f2{1.0001f, 1.0001f}, f2{0.5f, 0.25f}); x[2 * i] = v.x; x[2 * i + 1] = v.y; } }
            if (OP == 10) u[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1u, u[i]);
            if (OP == 11) { float r_; asm volatile("ds_bpermute_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r_) : "v"((int)(((threadIdx.x + 2) & 63) * 4)), "v"(x[i])); x[i] = r_; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; unsigned v = 0;
    for (int i = 0; i < 16; ++i) { s += x[i]; v += u[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + v;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
    const char *names[] = {"v_add_f32", "v_add_f32_dpp wave_shl:1", "v_mov_b32_dpp wave_shl:1", "v_add_f32_dpp row_shl:1", "v_sqrt_f32", "v_dot4_u32_u8", "v_sad_u32", "v_add+v_min_u32",
                           "v_fma_f32", "v_pk_fma_f32 (8 per 16 slots: 2 fma each)", "v_cvt_pk_u8_f32", "ds_bpermute_b32 + wait"};
    const int iters = 2000;
    for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2) {
        for (int op = 0; op < 12; ++op) {
            dim3 g(256), b(256 * wavesPerSimd);          // one workgroup per CU, 4 x wavesPerSimd waves
            switch (op) { case 0: k<0><<<g, b>>>(out, cyc, iters); break; case 1: k<1><<<g, b>>>(out, cyc, iters); break; case 2: k<2><<<g, b>>>(out, cyc, iters); break;
                          case 3: k<3><<<g, b>>>(out, cyc, iters); break; case 4: k<4><<<g, b>>>(out, cyc, iters); break; case 5: k<5><<<g, b>>>(out, cyc, iters); break;
                          case 6: k<6><<<g, b>>>(out, cyc, iters); break; case 7: k<7><<<g, b>>>(out, cyc, iters); break; case 8: k<8><<<g, b>>>(out, cyc, iters); break;
                          case 9: k<9><<<g, b>>>(out, cyc, iters); break; case 10: k<10><<<g, b>>>(out, cyc, iters); break; default: k<11><<<g, b>>>(out, cyc, iters); }
            hipDeviceSynchronize();
            // ... and the launch's wall time: every wave of the SIMD counted (wave 0 is the oldest and wins the issue arbitration)
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
            switch (op) { case 0: k<0><<<g, b>>>(out, cyc, iters); break; case 1: k<1><<<g, b>>>(out, cyc, iters); break; case 2: k<2><<<g, b>>>(out, cyc, iters); break;
                          case 3: k<3><<<g, b>>>(out, cyc, iters); break; case 4: k<4><<<g, b>>>(out, cyc, iters); break; case 5: k<5><<<g, b>>>(out, cyc, iters); break;
                          case 6: k<6><<<g, b>>>(out, cyc, iters); break; case 7: k<7><<<g, b>>>(out, cyc, iters); break; case 8: k<8><<<g, b>>>(out, cyc, iters); break;
                          case 9: k<9><<<g, b>>>(out, cyc, iters); break; case 10: k<10><<<g, b>>>(out, cyc, iters); break; default: k<11><<<g, b>>>(out, cyc, iters); }
            hipEventRecord(e1); hipEventSynchronize(e1); float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double perWave = iters * (op == 9 ? 8.0 : 16.0) * (op == 7 ? 2 : 1);
            printf("%d waves/SIMD  %-46s wave 0: %.2f clocks per instruction; SIMD: %.2f ns per wave-instruction (wall / instructions of one SIMD)\n", wavesPerSimd, names[op],
                   (double)h / perWave, ms * 1e6 / (perWave * wavesPerSimd));
        }
    }
    return 0;
}
