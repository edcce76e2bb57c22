// scale.hip -- Lanczos-3 resample, the MI355X-native replacement of shaders/scale.comp
// (reference: /root/reference/shaders/scale.comp:14-61, dispatched by src/scaler.cpp:260-395).
//
// The shader evaluates, per output pixel, 36 taps with 72 lanczos() calls (144 sin) and no reuse
// between neighbours.  Because its skip test is per axis (scale.comp:34-37) the filter is exactly
// separable, so here the per-axis tap indices and weights are tabulated once per (in,out) size on
// the host (lfg_capi.cpp: build_axis_table, same fp32 formula as the shader) and the kernels do
// only the weighted sums:  out = sum_y wy[y] * sum_x wx[x] * texel(sx+x, sy+y), weights
// pre-normalised, taps the shader skips carry weight 0.  Results are within +-1 LSB of the
// shader's own evaluation order (tests/test_gpu_parity.py).
//
// Roofline: HBM.  Algorithmic bytes = 4*(Win*Hin + Wout*Hout).
//
// Two kernels:
//   scale_2x_kernel      out == 2*in on both axes (every benchmark config).  One WAVE owns a strip
//                        of 64 input columns (= 128 output columns) and walks down the rows:
//                        coalesced uchar4 row loads -> LDS (one row, as float4) so each lane can
//                        read its 7 horizontal neighbours -> horizontal 6-tap sums for its two
//                        output columns -> a rolling six-row register window -> vertical 6-tap
//                        sums -> 8-byte coalesced stores.  No workgroup barrier: waves are
//                        independent, LDS is only the cross-lane exchange for the horizontal taps.
//   scale_generic_kernel any sizes (down-scaling included): one thread per output pixel, 36
//                        table-weighted texel reads through L1/L2.
#include <cstdio>

#include "lfg_device.hpp"
#include "lfg_internal.hpp"

namespace lfg {

// ------------------------------------------------------------------------------ generic

__global__ __launch_bounds__(256) void scale_generic_kernel(
    const uint8_t *__restrict__ in, int inW, int inH, int inPitch,
    uint8_t *__restrict__ out, int outW, int outH, int outPitch,
    const int *__restrict__ startX, const float *__restrict__ weightX,
    const int *__restrict__ startY, const float *__restrict__ weightY) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= outW || oy >= outH) return;
    const int sx = startX[ox], sy = startY[oy];
    float wx[6], wy[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { wx[k] = weightX[ox * 6 + k]; wy[k] = weightY[oy * 6 + k]; }
    float r = 0.f, g = 0.f, b = 0.f, a = 0.f;
#pragma unroll
    for (int y = 0; y < 6; ++y) {
        const int yy = clampi(sy + y, 0, inH - 1);
        const uint8_t *row = in + (size_t)yy * (size_t)inPitch;
        float hr = 0.f, hg = 0.f, hb = 0.f, ha = 0.f;
#pragma unroll
        for (int x = 0; x < 6; ++x) {
            const int xx = clampi(sx + x, 0, inW - 1);
            const uint32_t p = *reinterpret_cast<const uint32_t *>(row + (size_t)xx * 4u);
            hr = __builtin_fmaf(wx[x], byte0(p), hr);
            hg = __builtin_fmaf(wx[x], byte1(p), hg);
            hb = __builtin_fmaf(wx[x], byte2(p), hb);
            ha = __builtin_fmaf(wx[x], byte3(p), ha);
        }
        r = __builtin_fmaf(wy[y], hr, r);
        g = __builtin_fmaf(wy[y], hg, g);
        b = __builtin_fmaf(wy[y], hb, b);
        a = __builtin_fmaf(wy[y], ha, a);
    }
    *reinterpret_cast<uint32_t *>(out + (size_t)oy * (size_t)outPitch + (size_t)ox * 4u) =
        pack_rgba8_255(r, g, b, a);
}

hipError_t launch_scale_generic(hipStream_t s, const lfg_frame &in, const lfg_frame &out,
                                const AxisTable &tx, const AxisTable &ty) {
    dim3 grid((out.width + 63) / 64, (out.height + 3) / 4);
    hipLaunchKernelGGL(scale_generic_kernel, grid, dim3(256), 0, s,
                       (const uint8_t *)in.data, (int)in.width, (int)in.height, (int)in.pitch,
                       (uint8_t *)out.data, (int)out.width, (int)out.height, (int)out.pitch,
                       tx.d_start, tx.d_weight, ty.d_start, ty.d_weight);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ exact 2x

constexpr int kStripCols = 64;      // input columns per wave (one per lane)
constexpr int kRowBuf = 72;         // 64 + 3 left + 3 right halo, rounded up to 72 float4
constexpr int kOobOffset = (int)0x80000000;   // buffer offset that always fails the range check
#ifndef LFG_STORE_AUX
#define LFG_STORE_AUX 2
#endif
constexpr int kStoreAux = LFG_STORE_AUX;      // gfx940+ cache policy bits: 1 = sc0, 2 = nt, 16 = sc1

// Four channels as two register pairs: the weighted sums are written on pairs so that they compile to
// v_pk_fma_f32 (two FMAs per instruction).  Packed FMAs have no higher lane throughput than plain ones
// on gfx950, but a wave issues one VALU instruction per 4 cycles on its own, and this kernel is bound
// by per-wave issue latency, not by SIMD throughput: halving the instruction count of the FMA blocks
// shortens every step.
struct F4 { f32x2 lo, hi; };          // lo = (r, g), hi = (b, a)

__device__ __forceinline__ void fma4(F4 &acc, float w, const F4 &t) {
    const f32x2 ww = {w, w};
    acc.lo = __builtin_elementwise_fma(ww, t.lo, acc.lo);
    acc.hi = __builtin_elementwise_fma(ww, t.hi, acc.hi);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ F4 unpack255(uint32_t p) { return F4{f32x2{byte0(p), byte1(p)}, f32x2{byte2(p), byte3(p)}}; }

// At exact 2x, output columns 2k and 2k+1 (lane = input column k) read input columns k-3..k+2 and
// k-2..k+3; output rows 2r-5 and 2r-4 both read input rows r-5..r (host-verified: AxisTable::pattern_2x).
// Step r therefore loads input row r, forms the two horizontal sums, and emits those two output rows
// from the six most recent horizontal rows.  A wave runs STEPS emitting steps after five warm-up
// steps that fill the window.
//
// The whole strip is unrolled and BRANCH-FREE: every load and store is a raw buffer access whose
// offset is pushed out of range when it must not happen (lanes past the right edge, rows outside
// the image; the hardware drops out-of-range stores and returns 0 for loads).  Straight-line code
// lets the compiler count outstanding memory operations exactly, so each step waits only for the
// row it consumes (prefetched six steps earlier) instead of draining loads AND stores at every loop
// back-edge, which is what made the looped version latency-bound.
template <int STEPS>
__global__ __launch_bounds__(256, 3) void scale_2x_kernel(
    const uint8_t *__restrict__ in, int inW, int inH, int inPitch,
    uint8_t *__restrict__ out, int outW, int outH, int outPitch,
    const float *__restrict__ weightX, const float *__restrict__ weightY,
    int colGroups, int strips
#ifdef LFG_DIAG_STAMPS
    , unsigned long long *stamps     // diagnostic build only: per-wave {start, after prologue, end} s_memrealtime
#endif
    ) {
    constexpr int T = STEPS + 5;
#ifdef LFG_DIAG_STAMPS
    const unsigned long long tStart = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ __attribute__((aligned(16))) F4 rowbuf[4][kRowBuf];
    __shared__ __attribute__((aligned(16))) float wyS[4][STEPS * 2 * 6];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = (int)blockIdx.x * 4 + wv;              // global wave id, wave-uniform
    if (gw >= colGroups * strips) return;                 // no workgroup barrier below: safe
    const int cg = gw % colGroups, st = gw / colGroups;
    const int icx0 = cg * kStripCols;
    const int k = icx0 + lane;                            // this lane's input column
    // Emitting steps r = rBegin .. rBegin+STEPS-1 out of 2 .. inH+2; the last strip is shifted up
    // to full length (it re-emits a few rows of its neighbour with identical values).
    const int rBegin = min(2 + st * STEPS, inH + 3 - STEPS);
    const int rFirst = rBegin - 5;
    F4 *buf = rowbuf[wv];
    float *wys = wyS[wv];

    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(in), 0, inH * inPitch, 0x00020000);
    const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc(out, 0, outH * outPitch, 0x00020000);

    // Vertical weights of the strip's 2*STEPS output rows -> wave-private LDS once; the steps read them
    // with broadcast LDS reads (a scalar load per output row would sit on the critical path).
    {
        const int oyFirst = 2 * rBegin - 5;
        constexpr int kChunks = (STEPS * 12 + 63) / 64;
        float wv_[kChunks];
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {             // all loads first: one memory latency, not kChunks
            const int i = c * 64 + lane;
            const int oy = min(max(oyFirst + i / 6, 0), outH - 1);
            wv_[c] = weightY[(size_t)oy * 6u + (size_t)(i % 6)];
        }
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int i = c * 64 + lane;
            const int oy = oyFirst + i / 6;
            if (i < STEPS * 12) wys[i] = (oy >= 0 && oy < outH) ? wv_[c] : 0.0f;
        }
    }

    // Horizontal weights of the lane's two output columns: 12 consecutive floats, three 16-byte loads
    // (clamped for lanes past the right edge; those lanes never store).
    const int kc = min(k, inW - 1);
    float wa[6], wb[6];
    {
        const float4 *wp = reinterpret_cast<const float4 *>(weightX + (size_t)(2 * kc) * 6u);
        const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2];
        wa[0] = w0.x; wa[1] = w0.y; wa[2] = w0.z; wa[3] = w0.w; wa[4] = w1.x; wa[5] = w1.y;
        wb[0] = w1.z; wb[1] = w1.w; wb[2] = w2.x; wb[3] = w2.y; wb[4] = w2.z; wb[5] = w2.w;
    }

    // Row-buffer slot i holds input column icx0 - 3 + i: every lane loads slot `lane`, lanes 0..7 also
    // slot 64 + lane (the other lanes' second load is out of range and returns 0).
    const int offA = clampi(icx0 - 3 + lane, 0, inW - 1) * 4;
    const int offB = lane < kRowBuf - 64 ? clampi(icx0 - 3 + 64 + lane, 0, inW - 1) * 4 : kOobOffset;
    // Store offsets of the lane pair (2j, 2j+1): the even lane writes 16 bytes of the upper output row
    // of a step at column 4j, the odd lane 16 bytes of the lower row at the same column.  The scalar
    // part of the address is the upper row's offset.
    const bool odd = (lane & 1) != 0;
    const int offLane = k < inW ? (odd ? (k - 1) * 8 + outPitch : k * 8) : kOobOffset;
    auto load_row = [&](int r, uint32_t &pa, uint32_t &pb) {
        const int rowOff = clampi(r, 0, inH - 1) * inPitch;              // wave-uniform
        pa = __builtin_amdgcn_raw_buffer_load_b32(rIn, offA, rowOff, 0);
        pb = __builtin_amdgcn_raw_buffer_load_b32(rIn, offB, rowOff, 0);
    };

    constexpr int kAhead = 3;                              // rows in flight per wave
#ifdef LFG_DIAG_STAMPS
    const unsigned long long tPro = __builtin_amdgcn_s_memrealtime();
#endif
    F4 win[6][2];
    uint32_t pa[kAhead], pb[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; ++u) load_row(rFirst + u, pa[u], pb[u]);

#pragma unroll
    for (int s = 0; s < T; ++s) {
        const int r = rFirst + s;
        const int u = s % 6;                               // window slot of this step
        // 1. this row's pixels -> LDS as floats (0..255 scale); request the row kAhead steps ahead.
        const int v = s % kAhead;
        buf[lane] = unpack255(pa[v]);
        if (lane < kRowBuf - 64) buf[64 + lane] = unpack255(pb[v]);
        if (s + kAhead < T) load_row(r + kAhead, pa[v], pb[v]);
        wave_lds_sync();
        // 2. the lane's 7 horizontal neighbours (input columns k-3 .. k+3).
        F4 t[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) t[j] = buf[lane + j];
        wave_lds_sync();
        F4 h0{f32x2{0, 0}, f32x2{0, 0}}, h1{f32x2{0, 0}, f32x2{0, 0}};
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            fma4(h0, wa[j], t[j]);
            fma4(h1, wb[j], t[j + 1]);
        }
        win[u][0] = h0; win[u][1] = h1;
        // 3. emit output rows 2r-5 and 2r-4 from window rows r-5..r = slots (u+1+j)%6.
        if (s >= 5) {
            uint32_t px[2][2];                                         // [output row half][column]
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const float2 *wyp = reinterpret_cast<const float2 *>(wys + (2 * (s - 5) + half) * 6);
                const float2 wy01 = wyp[0], wy23 = wyp[1], wy45 = wyp[2];
                const float wy[6] = {wy01.x, wy01.y, wy23.x, wy23.y, wy45.x, wy45.y};
                F4 o0{f32x2{0, 0}, f32x2{0, 0}}, o1{f32x2{0, 0}, f32x2{0, 0}};
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    fma4(o0, wy[j], win[(u + 1 + j) % 6][0]);
                    fma4(o1, wy[j], win[(u + 1 + j) % 6][1]);
                }
                px[half][0] = pack_rgba8_255(o0.lo.x, o0.lo.y, o0.hi.x, o0.hi.y);
                px[half][1] = pack_rgba8_255(o1.lo.x, o1.lo.y, o1.hi.x, o1.hi.y);
            }
            // One 16-byte store per lane instead of two 8-byte ones (8-byte-per-lane stores are
            // issue-bound at ~7 B/clk/CU on gfx950, which capped this kernel at ~7.5 us).  Lanes pair up:
            // the even lane takes both lanes' pixels of the upper row (4 adjacent columns), the odd
            // lane both lanes' pixels of the lower row.  The exchange is one DPP quad_perm [1,0,3,2]
            // (swap neighbours) per dword of the pair each lane gives away.
            const uint32_t g0 = odd ? px[0][0] : px[1][0], g1 = odd ? px[0][1] : px[1][1];
            const uint32_t x0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g0, 0xB1, 0xF, 0xF, false);
            const uint32_t x1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)g1, 0xB1, 0xF, 0xF, false);
            u32x4 q;
            q.x = odd ? x0 : px[0][0]; q.y = odd ? x1 : px[0][1];
            q.z = odd ? px[1][0] : x0; q.w = odd ? px[1][1] : x1;
            const int oy0 = 2 * r - 5;                                 // upper row (wave-uniform), >= -1
            // Only two steps of a frame are special: oy0 == -1 (no upper row; the lower row is row 0, so
            // the odd lanes drop the extra pitch) and oy0 + 1 == outH (no lower row).
            const int sub = oy0 < 0 ? outPitch : 0;                    // scalar
            const bool ok = odd ? (oy0 + 1 < outH) : (oy0 >= 0);
            const int voff = ok ? offLane - sub : kOobOffset;
            __builtin_amdgcn_raw_buffer_store_b128(q, rOut, voff, max(oy0, 0) * outPitch, kStoreAux);
            // A buffer store of more than 8 bytes reads its data registers over several cycles after issue; a VALU
            // write to them in the next slots wins the race (seen on gfx950 as the next row's unpacked float in
            // the first pixel of lanes 12/14 mod 16, a few dozen pixels per 4K frame, not every run).  The
            // compiler's hazard recogniser leaves the case "scalar register in the soffset field" out, so the
            // wait states are spelled out here.
            asm volatile("s_nop 7");
        }
        __builtin_amdgcn_sched_barrier(0);     // keep each step's registers local
    }
#ifdef LFG_DIAG_STAMPS
    if (stamps && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[(size_t)gw * 4 + 0] = tStart; stamps[(size_t)gw * 4 + 1] = tPro;
        stamps[(size_t)gw * 4 + 2] = __builtin_amdgcn_s_memrealtime();
        stamps[(size_t)gw * 4 + 3] = 0;
    }
#endif
}

constexpr int kScaleSteps = 11;

#ifdef LFG_DIAG_STAMPS
// Diagnostic build only (-DLFG_DIAG_STAMPS, never the shipped library): a device buffer for the
// per-wave time stamps, dumped to $LFG_STAMPS_FILE by lfg_diag_dump_stamps().
static unsigned long long *g_stamps = nullptr;
static size_t g_stampWaves = 0;
static unsigned long long *diag_stamp_buffer(size_t waves) {
    if (!g_stamps) { (void)hipMalloc((void **)&g_stamps, waves * 4 * sizeof(unsigned long long)); g_stampWaves = waves; }
    return g_stamps;
}
extern "C" __attribute__((visibility("default"))) int lfg_diag_dump_stamps(const char *path) {
    if (!g_stamps) return -1;
    std::vector<unsigned long long> h(g_stampWaves * 4);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "w");
    if (!f) return -2;
    for (size_t i = 0; i < g_stampWaves; ++i) fprintf(f, "%zu %llu %llu %llu\n", i, h[4 * i], h[4 * i + 1], h[4 * i + 2]);
    fclose(f);
    return 0;
}
#endif

bool scale_2x_supported(const lfg_frame &in, const lfg_frame &out) {
    return (int)in.height + 1 >= kScaleSteps && in.width % 2u == 0 && out.pitch % 16u == 0 &&
           (uintptr_t)out.data % 16u == 0 && (uint64_t)in.height * in.pitch < 0x7fffffffull &&
           (uint64_t)out.height * out.pitch < 0x7fffffffull;
}

hipError_t launch_scale_2x(hipStream_t s, const lfg_frame &in, const lfg_frame &out,
                           const AxisTable &tx, const AxisTable &ty) {
    const int colGroups = ((int)in.width + kStripCols - 1) / kStripCols;
    const int totalSteps = (int)in.height + 1;            // steps r = 2 .. inH + 2
    const int strips = (totalSteps + kScaleSteps - 1) / kScaleSteps;
    const int waves = colGroups * strips;
    dim3 grid((waves + 3) / 4);
    hipLaunchKernelGGL(scale_2x_kernel<kScaleSteps>, grid, dim3(256), 0, s,
                       (const uint8_t *)in.data, (int)in.width, (int)in.height, (int)in.pitch,
                       (uint8_t *)out.data, (int)out.width, (int)out.height, (int)out.pitch,
                       tx.d_weight, ty.d_weight, colGroups, strips
#ifdef LFG_DIAG_STAMPS
                       , diag_stamp_buffer((size_t)waves)
#endif
                       );
    return hipGetLastError();
}

}  // namespace lfg
