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
#include <cstdlib>

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

struct F4 { float x, y, z, w; };

__device__ __forceinline__ F4 unpack255(uint32_t p) { return F4{byte0(p), byte1(p), byte2(p), byte3(p)}; }

// At exact 2x, output columns 2k and 2k+1 (lane = input column k) read input columns k-3..k+2 and
// k-2..k+3; output rows 2r-5 and 2r-4 both read input rows r-5..r (host-verified: AxisTable::pattern_2x).
// Step r therefore loads input row r, forms the two horizontal sums, and emits those two output rows
// from the six most recent horizontal rows.  Strip `st` runs steps [2 + st*stepsPerStrip, ...), after
// five warm-up steps that refill the window.
__global__ __launch_bounds__(256) void scale_2x_kernel(
    const uint8_t *__restrict__ in, int inW, int inH, int inPitch,
    uint8_t *__restrict__ out, int outW, int outH, int outPitch,
    const float *__restrict__ weightX, const float *__restrict__ weightY,
    int colGroups, int strips, int stepsPerStrip) {
    __shared__ __attribute__((aligned(16))) F4 rowbuf[4][kRowBuf];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int gw = (int)blockIdx.x * 4 + wv;              // global wave id, wave-uniform
    if (gw >= colGroups * strips) return;                 // no workgroup barrier below: safe
    const int cg = gw % colGroups, st = gw / colGroups;
    const int icx0 = cg * kStripCols;
    const int k = icx0 + lane;                            // this lane's input column
    const int rEnd = min(2 + (st + 1) * stepsPerStrip, inH + 3);
    const int rBegin = 2 + st * stepsPerStrip;            // first emitting step
    F4 *buf = rowbuf[wv];

    // Horizontal weights of the lane's two output columns (clamped for lanes past the right edge;
    // those lanes never store).
    const int kc = min(k, inW - 1);
    float wa[6], wb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        wa[j] = weightX[(2 * kc) * 6 + j];
        wb[j] = weightX[(2 * kc + 1) * 6 + j];
    }

    // Column this lane loads into the row buffer: buffer slot i holds input column icx0 - 3 + i.
    const int colA = clampi(icx0 - 3 + lane, 0, inW - 1);
    const int colB = clampi(icx0 - 3 + 64 + lane, 0, inW - 1);    // lanes 0..7 only
    auto load_row = [&](int r, uint32_t &pa, uint32_t &pb) {
        const int rr = clampi(r, 0, inH - 1);
        const uint8_t *row = in + (size_t)rr * (size_t)inPitch;
        pa = *reinterpret_cast<const uint32_t *>(row + (size_t)colA * 4u);
        pb = 0u;
        if (lane < kRowBuf - 64) pb = *reinterpret_cast<const uint32_t *>(row + (size_t)colB * 4u);
    };

    F4 win[6][2];
#pragma unroll
    for (int u = 0; u < 6; ++u) { win[u][0] = F4{0, 0, 0, 0}; win[u][1] = F4{0, 0, 0, 0}; }

    // Software pipeline over rows: six row loads are always in flight per wave (the row for step r+6
    // is requested as soon as step r has consumed its registers), so HBM/L2 latency is paid once per
    // wave instead of once per row.
    const int rFirst = rBegin - 5;
    uint32_t pa[6], pb[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) load_row(rFirst + u, pa[u], pb[u]);

    for (int rb = rFirst; rb < rEnd; rb += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int r = rb + u;
            if (r < rEnd) {
                // 1. this row's pixels -> LDS as floats (0..255 scale).
                buf[lane] = unpack255(pa[u]);
                if (lane < kRowBuf - 64) buf[64 + lane] = unpack255(pb[u]);
                load_row(r + 6, pa[u], pb[u]);
                wave_lds_sync();
                // 2. the lane's 7 horizontal neighbours (input columns k-3 .. k+3).
                F4 t[7];
#pragma unroll
                for (int j = 0; j < 7; ++j) t[j] = buf[lane + j];
                wave_lds_sync();
                F4 h0{0, 0, 0, 0}, h1{0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    h0.x = __builtin_fmaf(wa[j], t[j].x, h0.x); h0.y = __builtin_fmaf(wa[j], t[j].y, h0.y);
                    h0.z = __builtin_fmaf(wa[j], t[j].z, h0.z); h0.w = __builtin_fmaf(wa[j], t[j].w, h0.w);
                    h1.x = __builtin_fmaf(wb[j], t[j + 1].x, h1.x); h1.y = __builtin_fmaf(wb[j], t[j + 1].y, h1.y);
                    h1.z = __builtin_fmaf(wb[j], t[j + 1].z, h1.z); h1.w = __builtin_fmaf(wb[j], t[j + 1].w, h1.w);
                }
                win[u][0] = h0; win[u][1] = h1;
                // 3. emit output rows 2r-5 and 2r-4 from window rows r-5..r = slots (u+1+j)%6.
                if (r >= rBegin) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int oy = 2 * r - 5 + half;              // wave-uniform
                        if (oy >= 0 && oy < outH) {
                            const float *wy = weightY + (size_t)oy * 6u;
                            F4 o0{0, 0, 0, 0}, o1{0, 0, 0, 0};
#pragma unroll
                            for (int j = 0; j < 6; ++j) {
                                const float w = wy[j];
                                const F4 &a = win[(u + 1 + j) % 6][0];
                                const F4 &b = win[(u + 1 + j) % 6][1];
                                o0.x = __builtin_fmaf(w, a.x, o0.x); o0.y = __builtin_fmaf(w, a.y, o0.y);
                                o0.z = __builtin_fmaf(w, a.z, o0.z); o0.w = __builtin_fmaf(w, a.w, o0.w);
                                o1.x = __builtin_fmaf(w, b.x, o1.x); o1.y = __builtin_fmaf(w, b.y, o1.y);
                                o1.z = __builtin_fmaf(w, b.z, o1.z); o1.w = __builtin_fmaf(w, b.w, o1.w);
                            }
                            if (k < inW) {
                                uint2 px;
                                px.x = pack_rgba8_255(o0.x, o0.y, o0.z, o0.w);
                                px.y = pack_rgba8_255(o1.x, o1.y, o1.z, o1.w);
                                *reinterpret_cast<uint2 *>(out + (size_t)oy * (size_t)outPitch + (size_t)k * 8u) = px;
                            }
                        }
                    }
                }
            }
        }
    }
}

static int scale2x_steps_per_strip() {
    static int v = [] {
        const char *e = getenv("LFG_SCALE_STEPS");       // tuning knob for experiments only
        int n = e ? atoi(e) : 0;
        return (n >= 1 && n <= 4096) ? n : 8;
    }();
    return v;
}

hipError_t launch_scale_2x(hipStream_t s, const lfg_frame &in, const lfg_frame &out,
                           const AxisTable &tx, const AxisTable &ty) {
    const int colGroups = ((int)in.width + kStripCols - 1) / kStripCols;
    const int totalSteps = (int)in.height + 1;            // steps r = 2 .. inH + 2
    const int stepsPerStrip = scale2x_steps_per_strip();
    const int strips = (totalSteps + stepsPerStrip - 1) / stepsPerStrip;
    const int waves = colGroups * strips;
    dim3 grid((waves + 3) / 4);
    hipLaunchKernelGGL(scale_2x_kernel, grid, dim3(256), 0, s,
                       (const uint8_t *)in.data, (int)in.width, (int)in.height, (int)in.pitch,
                       (uint8_t *)out.data, (int)out.width, (int)out.height, (int)out.pitch,
                       tx.d_weight, ty.d_weight, colGroups, strips, stepsPerStrip);
    return hipGetLastError();
}

}  // namespace lfg
