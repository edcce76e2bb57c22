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
//   scale_2x_kernel      out == 2*in on both axes (every benchmark config).  One WAVE owns a strip of 120 input
//                        columns (= 240 output columns, two input columns per lane) and walks down a few rows:
//                        format-converting row loads -> a rolling six-row register window of floats -> vertical
//                        6-tap sums -> wave-private LDS exchange with the neighbouring lanes -> horizontal 6-tap
//                        sums -> 16-byte coalesced stores.  No workgroup barrier: waves are independent.
//   scale_generic_kernel any sizes (down-scaling included): one thread per output pixel, 36
//                        table-weighted texel reads through L1/L2.
#include <algorithm>
#include <cstdio>
#include <vector>

#include "lfg_device.hpp"
#include "lfg_internal.hpp"
#include "lfg_interp.hpp"

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
//
// Work decomposition.  At exact 2x, output rows 2r-5 and 2r-4 both read input rows r-5..r, and
// output columns 2c+1 and 2c+2 both read input columns c-2..c+3 (host-verified: AxisTable::pattern_2x).
//
//   lane   owns TWO adjacent input columns c0 = B-4+2L, c1 = c0+1 and therefore FOUR adjacent output columns
//          2c0..2c0+3 = one 16-byte store per output row, no lane exchange for the store;
//   wave   64 lanes = 128 input columns of which the middle 120 (lanes 2..61) are owned: the two lanes on either side
//          only feed their columns to the horizontal taps of their neighbours (3 columns of halo per side);
//   strip  S emitting steps; step r loads input row r, emits output rows 2r-5 and 2r-4.
//
// VERTICAL FIRST.  The rolling window holds the six most recent INPUT rows of the lane's two columns as floats (they
// arrive as floats: buffer_load_format_xyzw through an 8_8_8_8 USCALED descriptor converts in the texture-address
// unit, no v_cvt_f32_ubyte).  A step forms the vertical 6-tap sums of its two output rows at the lane's two input
// columns (48 packed FMAs, weights wave-uniform), hands them to the neighbouring lanes through a wave-private LDS
// slab (4 x ds_write_b128, 12 x ds_read_b128, no workgroup barrier: one wave's DS operations execute in order), and
// finishes with the horizontal 6-tap sums of its 2 x 4 output pixels (96 packed FMAs), v_cvt_pk_u8_f32 (round half
// to even + saturate) and two 16-byte stores.  Because the window holds raw input rows, the five rows a strip needs
// before its first output cost five row loads and NO arithmetic -- strips can be short, a frame becomes several
// thousand waves in more than one generation, and a wave that waits for its first rows shares its SIMD with waves
// that are computing.  (The previous kernel filtered horizontally first: every strip recomputed five rows of
// horizontal sums, so strips had to be long, all waves started together and spent the first 3.5 us waiting.)
//
// Cost per step and wave: 144 v_pk_fma_f32 + 32 v_cvt_pk_u8_f32 + ~10 other VALU for 2 x 240 owned output pixels =
// 97 VALU cycles per 64 output pixels against a floor of 88 (36 FMAs + 4 conversions per pixel), 6.7 % of it the halo lanes.
//
// Everything is straight-line and branch-free per strip (out-of-range buffer offsets instead of predicates), so the
// compiler's vmcnt bookkeeping is exact and each step waits only for the row it consumes.

constexpr int kOwnedLanes = 60;                       // lanes 2..61
constexpr int kOwnedCols = 2 * kOwnedLanes;           // input columns owned per wave
constexpr int kOobOffset = (int)0x80000000;           // buffer offset that always fails the range check
#ifndef LFG_STORE_AUX
#define LFG_STORE_AUX 2
#endif
constexpr int kStoreAux = LFG_STORE_AUX;              // gfx940+ cache policy bits: 1 = sc0, 2 = nt, 16 = sc1
// Strip lengths (emitting steps) of the three dispatch layers, see scale_2x_strip_plan().
#ifndef LFG_SCALE_L0
#define LFG_SCALE_L0 7
#endif
#ifndef LFG_SCALE_L1
#define LFG_SCALE_L1 6
#endif
#ifndef LFG_SCALE_L2
#define LFG_SCALE_L2 4
#endif
#define LFG_SCALE_STEPS LFG_SCALE_L0                  // the longest strip: what the kernel is unrolled to
#ifndef LFG_SCALE_STAGGER
#define LFG_SCALE_STAGGER 0                           // x 64 clocks between the first loads of the waves sharing a SIMD
#endif
#ifndef LFG_SCALE_AHEAD
#define LFG_SCALE_AHEAD 1                             // input rows requested beyond the six of the current step
#endif
#ifndef LFG_SCALE_WAVES
#define LFG_SCALE_WAVES 3                             // waves per SIMD the register allocation aims at
#endif

// Four channels as two register pairs, so the weighted sums compile to v_pk_fma_f32 / v_pk_mul_f32: the same lane
// throughput as plain FMAs on gfx950 but half the issue slots, which the LDS and memory instructions need.
struct F4 { f32x2 lo, hi; };          // lo = (r, g), hi = (b, a)

__device__ __forceinline__ F4 to_f4(f32x4 v) { return F4{f32x2{v.x, v.y}, f32x2{v.z, v.w}}; }
__device__ __forceinline__ f32x4 to_v4(const F4 &t) { return f32x4{t.lo.x, t.lo.y, t.hi.x, t.hi.y}; }
__device__ __forceinline__ F4 mul4(float w, const F4 &t) {
    const f32x2 ww = {w, w};
    return F4{ww * t.lo, ww * t.hi};
}
__device__ __forceinline__ void fma4(F4 &acc, float w, const F4 &t) {
    const f32x2 ww = {w, w};
    acc.lo = __builtin_elementwise_fma(ww, t.lo, acc.lo);
    acc.hi = __builtin_elementwise_fma(ww, t.hi, acc.hi);
}
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Strips of the 2x kernel for an input of inH rows.  The inH + 1 steps r = 2 .. inH + 2 are cut into eight contiguous
// bands, one per XCD; a band into stripsPerXcd strips of three lengths, longest first, in the order their workgroups
// are dispatched: strip i of XCD x starts at step `first` and has `steps` steps (0: nothing left of the band).
// Why three lengths: the waves of a launch all start within half a microsecond and every one first needs six input
// rows; the three workgroups a CU holds get going about 1.7 us apart (per-wave stamps: first vertical pass done at
// 1.8 / 3.5 / 5.5 us -- the oldest wave of a SIMD wins the issue arbitration), so with equal strips the workgroups
// that started last also finish last, alone on their SIMDs; with L0 > L1 > L2 the three finish together.  Nothing
// depends on the dispatch order but the run time.  Host and device evaluate the same function.
__host__ __device__ inline void scale_2x_strip_of(int inH, int x, int i, int stripsPerXcd, int &first, int &steps) {
    const int total = inH + 1;
    const int b0 = 2 + (total * x) / 8, b1 = 2 + (total * (x + 1)) / 8;
    const int k = stripsPerXcd / 3;
    const int i0 = i < k ? i : k, i1 = i < k ? 0 : (i < 2 * k ? i - k : k), i2 = i < 2 * k ? 0 : i - 2 * k;
    first = b0 + i0 * LFG_SCALE_L0 + i1 * LFG_SCALE_L1 + i2 * LFG_SCALE_L2;
    const int len = i < k ? LFG_SCALE_L0 : (i < 2 * k ? LFG_SCALE_L1 : LFG_SCALE_L2);
    const int left = b1 - first;
    steps = left < len ? (left > 0 ? left : 0) : len;
}

typedef f32x4 (*SlabPtr)[2][2][68];                    // [step parity][row A | row B][even | odd column][lane + 2]

// Where a strip's input rows come from.  Plain: the input frame `in`.  Fused (the
// input-resolution data flow): there is no input frame -- a row is interpolated on the fly from prev, curr and the
// motion vectors (all at input resolution) with interpolate.comp's own arithmetic, rounded to bytes exactly as the
// interpolate stage stores it, and upscaled at once: the generated frame never exists at input resolution in memory.
struct FusedSource {
    const uint8_t *prev; int prevPitch;
    const uint8_t *curr; int currPitch;
    const int8_t *mv; int mvPitch;
    float t; int intended;
};

// One strip: STEPS unrolled steps of which the first n emit (n <= STEPS; the others' stores are switched off).
template <int STEPS, bool FUSED>
__device__ __forceinline__ void scale_2x_strip(
    const FusedSource &fs, const uint8_t *__restrict__ in, int inW, int inH, int inPitch,
    uint8_t *__restrict__ out, int outW, int outH, int outPitch,
    const uint8_t *__restrict__ classX, const float *__restrict__ paletteX, const float *__restrict__ weightY,
    int rBegin, int n, int cg, int laneArg, SlabPtr slabW
#ifdef LFG_DIAG_STAMPS
    , unsigned long long *stamps, unsigned long long tStart, unsigned long long clk0, int st, int wv
#endif
    ) {
#ifdef LFG_DIAG_STAMPS
    unsigned long long tS[LFG_SCALE_STEPS + 2] = {};
    tS[0] = tStart;
#endif
    // (a strip reads n + 5 input rows; the unrolled sequence requests rows as if it had STEPS steps)
    constexpr int T = STEPS + 5;
    // The lane number is taken afresh in every body (volatile: not merged across the three): kept alive from the kernel's
    // entry across the three-way branch it was spilled, and the shortest body's reload -- s_waitcnt vmcnt(0) -- sat between
    // its first two row loads and the rest.
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(lane));
    (void)laneArg;
    const int c0 = cg * kOwnedCols - 4 + 2 * lane;        // the lane's even input column (c1 = c0 + 1)
    const bool owned = lane >= 2 && lane < 2 + kOwnedLanes && c0 < inW;
    const int rFirst = rBegin - 5;

    const __amdgpu_buffer_rsrc_t rIn = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint8_t *>(in), 0, inH * inPitch, kRsrcRgba8Uscaled);
    const __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc(out, 0, outH * outPitch, kRsrcRaw32);

    // Input columns outside the image carry weight 0 in every tap that reaches them; they only have to load something
    // finite, and an out-of-range offset loads 0.  (Rows are clamped instead: the scalar row offset is not part of
    // the hardware's range check.)
    const int voff0 = (c0 >= 0 && c0 < inW) ? c0 * 4 : kOobOffset;
    const int voff1 = (c0 + 1 >= 0 && c0 + 1 < inW) ? (c0 + 1) * 4 : kOobOffset;
    const uint32_t cls = *reinterpret_cast<const uint32_t *>(classX + 2 * clampi(c0, 0, inW - 2));   // lanes outside the image never store
    auto load_row = [&](int r, F4 &a, F4 &b) {
        if (FUSED) {
            // the lane's two pixels of the interpolated row (rows clamped like the loads of the plain kernel; columns
            // outside the image carry weight 0 everywhere: any finite value will do)
            const int y = clampi(r, 0, inH - 1);
            uint32_t px[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int x = clampi(c0 + c, 0, inW - 1);
                const int8_t *m = fs.mv + (size_t)y * (size_t)fs.mvPitch + (size_t)x * 2u;
                float mx = (float)m[0], my = (float)m[1];
                if (fs.intended) { mx = mx / (float)inW; my = my / (float)inH; }
                px[c] = interpolate_pixel(fs.prev, fs.prevPitch, fs.curr, fs.currPitch, inW, inH, x, y, mx, my, fs.t);
            }
            a = F4{f32x2{byte0(px[0]), byte1(px[0])}, f32x2{byte2(px[0]), byte3(px[0])}};
            b = F4{f32x2{byte0(px[1]), byte1(px[1])}, f32x2{byte2(px[1]), byte3(px[1])}};
            return;
        }
        const int rowOff = clampi(r, 0, inH - 1) * inPitch;                 // wave-uniform
        a = to_f4(buffer_load_rgba8_format(rIn, voff0, rowOff));
        b = to_f4(buffer_load_rgba8_format(rIn, voff1, rowOff));
    };

    // The window: six input rows x two columns, plus LFG_SCALE_AHEAD rows in flight behind them.  Everything the
    // first steps need is requested before anything else.
    constexpr int W = 6 + LFG_SCALE_AHEAD;
    F4 win[W][2];
#pragma unroll
    for (int t = 0; t < W && t < T; ++t) load_row(rFirst + t, win[t][0], win[t][1]);
    // called once the vertical pass of step s has consumed row s: its slot takes row s + W.
    auto advance = [&](int s) {
        if (s + W < T) load_row(rFirst + s + W, win[s % W][0], win[s % W][1]);
    };

    // Horizontal weights of the lane's four output columns 2c0 .. 2c0+3: one class byte per column (a 4-byte load,
    // requested before anything else: its answer is the address of the next loads), then the class's row of the
    // palette -- 31 distinct rows at 1920 -> 3840, one of them for 97 % of the columns, so nearly every lane of every
    // wave reads the same 32 bytes (L1) instead of its own 96 bytes of table.
    float wx[4][6];
    {
        float4 lo[4]; float2 hi[4];
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            const float *row = paletteX + ((cls >> (8 * oc)) & 0xffu) * 8u;
            lo[oc] = *reinterpret_cast<const float4 *>(row);
            hi[oc] = *reinterpret_cast<const float2 *>(row + 4);
        }
#pragma unroll
        for (int oc = 0; oc < 4; ++oc) {
            wx[oc][0] = lo[oc].x; wx[oc][1] = lo[oc].y; wx[oc][2] = lo[oc].z; wx[oc][3] = lo[oc].w;
            wx[oc][4] = hi[oc].x; wx[oc][5] = hi[oc].y;
        }
    }

    const int offLane = owned ? c0 * 8 : kOobOffset;       // byte offset of output column 2 * c0

    // Vertical weights are wave-uniform: scalar loads, twelve floats per step (six per output row), requested a step
    // ahead (scalar loads return out of order with LDS traffic, so a wait while one is in flight is a full lgkmcnt(0)).
    auto load_wy = [&](int r, float (&a)[6], float (&b)[6]) {
        const float *pa = weightY + (size_t)clampi(2 * r - 5, 0, outH - 1) * 6u;
        const float *pb = weightY + (size_t)clampi(2 * r - 4, 0, outH - 1) * 6u;
#pragma unroll
        for (int j = 0; j < 6; ++j) { a[j] = pa[j]; b[j] = pb[j]; }
    };
    // Vertical 6-tap sums of step s (output rows 2r-5 | 2r-4) at the lane's two input columns.  Window rows r-5 .. r
    // are the strip's rows s .. s+5, held in slots (s + j) % W.
    auto vertical = [&](int s, const float (&wyA)[6], const float (&wyB)[6], F4 (&vA)[2], F4 (&vB)[2]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#ifdef LFG_DIAG_NO_MATH        // timing experiments only: the loads stay alive, the arithmetic goes
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("" : : "v"(win[(s + j) % W][c].lo), "v"(win[(s + j) % W][c].hi));
            vA[c] = mul4(wyA[0], win[s % W][c]); vB[c] = mul4(wyB[5], win[(s + 5) % W][c]);
#else
            vA[c] = mul4(wyA[0], win[s % W][c]);
            vB[c] = mul4(wyB[0], win[s % W][c]);
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                fma4(vA[c], wyA[j], win[(s + j) % W][c]);
                fma4(vB[c], wyB[j], win[(s + j) % W][c]);
            }
#endif
        }
    };
    // Exchange slab of step s: [step parity][row A | row B][even | odd column][lane + 2].
    auto slab_of = [&](int s, int half, int odd) -> f32x4 * { return &slabW[s & 1][half][odd][2]; };
    auto publish = [&](int s, const F4 (&vA)[2], const F4 (&vB)[2]) {
        slab_of(s, 0, 0)[lane] = to_v4(vA[0]); slab_of(s, 0, 1)[lane] = to_v4(vA[1]);
        slab_of(s, 1, 0)[lane] = to_v4(vB[0]); slab_of(s, 1, 1)[lane] = to_v4(vB[1]);
    };

    // Software pipeline: while the sums of step s travel through the slab, the vertical pass of step s+1 runs; the
    // two slabs alternate, so nothing waits for a write to be read back except the very first step.
    float wy[2][2][6];                                     // [step parity][row A | row B][tap]
    F4 v[2][2][2];                                         // [step parity][row A | row B][column]
    load_wy(rBegin, wy[0][0], wy[0][1]);
    if (STEPS > 1) load_wy(rBegin + 1, wy[1][0], wy[1][1]);
    vertical(0, wy[0][0], wy[0][1], v[0][0], v[0][1]);
#ifdef LFG_DIAG_STAMPS
    asm volatile("" : "+v"(v[0][0][0].lo), "+v"(v[0][1][1].hi)); tS[1] = __builtin_amdgcn_s_memrealtime();
#endif
    advance(0);                                            // the oldest row is dead: its slot takes a later row
    publish(0, v[0][0], v[0][1]);

#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int r = rBegin + s;
        const int oyA = 2 * r - 5, oyB = 2 * r - 4;        // wave-uniform; oyA >= -1, oyB <= outH
        const int p = s & 1, q = p ^ 1;
        wave_lds_sync();
        // neighbours' sums of row A of this step: input columns c0-3 .. c0+4 are
        //   o[L-2] e[L-1] o[L-1] | own e, own o | e[L+1] o[L+1] e[L+2]
        F4 x[2][8];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const f32x4 *se = slab_of(s, half, 0), *so = slab_of(s, half, 1);
            x[half][0] = to_f4(so[lane - 2]); x[half][1] = to_f4(se[lane - 1]); x[half][2] = to_f4(so[lane - 1]);
            x[half][3] = v[p][half][0]; x[half][4] = v[p][half][1];
            x[half][5] = to_f4(se[lane + 1]); x[half][6] = to_f4(so[lane + 1]); x[half][7] = to_f4(se[lane + 2]);
        }
        // the next step's vertical pass, its row request and its hand-over, while those reads are in flight
        if (s + 1 < STEPS) {                               // (also for the last step of a shorter strip: cheaper than a branch)
            vertical(s + 1, wy[q][0], wy[q][1], v[q][0], v[q][1]);
            advance(s + 1);
            publish(s + 1, v[q][0], v[q][1]);
            if (s + 2 < STEPS) load_wy(r + 2, wy[p][0], wy[p][1]);
        }
        // horizontal 6-tap sums, rounding, store
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            uint32_t px[4];
#pragma unroll
            for (int oc = 0; oc < 4; ++oc) {
                const int first = (oc + 1) >> 1;           // output column 2c0+oc reads x[first .. first+5]
                F4 o = mul4(wx[oc][0], x[half][first]);
#ifdef LFG_DIAG_NO_MATH
#pragma unroll
                for (int j = 1; j < 6; ++j) asm volatile("" : : "v"(x[half][first + j].lo), "v"(x[half][first + j].hi), "v"(wx[oc][j]));
#else
#pragma unroll
                for (int j = 1; j < 6; ++j) fma4(o, wx[oc][j], x[half][first + j]);
#endif
                px[oc] = pack_rgba8_255(o.lo.x, o.lo.y, o.hi.x, o.hi.y);
            }
            const int oy = half ? oyB : oyA;
            const bool rowOk = s < n && (half ? (oyB < outH) : (oyA >= 0));      // scalar
            const int voff = rowOk ? offLane + oy * outPitch : kOobOffset;
#ifdef LFG_DIAG_NO_STORE
            asm volatile("" : : "v"(px[0]), "v"(px[1]), "v"(px[2]), "v"(px[3]), "v"(voff));
#else
            store_b128_guarded<kStoreAux>(u32x4{px[0], px[1], px[2], px[3]}, rOut, voff);
#endif
        }
#ifdef LFG_DIAG_STAMPS
        tS[2 + s] = __builtin_amdgcn_s_memrealtime();
#endif
    }
#ifdef LFG_DIAG_STAMPS
    if (stamps && lane == 0) {
        const size_t wid = ((size_t)blockIdx.x * 4 + wv) * (LFG_SCALE_STEPS + 4);
        unsigned hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        for (int i = 0; i < LFG_SCALE_STEPS + 2; ++i) stamps[wid + i] = i < n + 2 ? tS[i] : tS[n + 1];
        stamps[wid + LFG_SCALE_STEPS + 2] = ((unsigned long long)xcc << 32) | hw;
        // shader clocks over the wave's life (s_memtime) in the upper half: clock = that / (end - start) x 100 MHz
        stamps[wid + LFG_SCALE_STEPS + 3] = ((__builtin_amdgcn_s_memtime() - clk0) << 32) | ((unsigned)n << 24) | ((unsigned)st << 8) | (unsigned)cg;
    }
#endif
}

// (the fused variant carries interpolate.comp's sampling code next to the filter: two waves per SIMD instead of three)
template <bool FUSED>
__global__ __launch_bounds__(256, FUSED ? 2 : LFG_SCALE_WAVES) void scale_2x_kernel(
    FusedSource fs, const uint8_t *__restrict__ in, int inW, int inH, int inPitch,
    uint8_t *__restrict__ out, int outW, int outH, int outPitch,
    const uint8_t *__restrict__ classX, const float *__restrict__ paletteX, const float *__restrict__ weightY,
    int colGroups, int stripsPerXcd
#ifdef LFG_DIAG_STAMPS
    , unsigned long long *stamps     // diagnostic build only: per wave {start, first rows in, each step's end} s_memrealtime + hw id
#endif
    ) {
#ifdef LFG_DIAG_STAMPS
    const unsigned long long tStart = __builtin_amdgcn_s_memrealtime();
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    // Wave-private exchange slabs: [wave][step parity][row A | row B][even column of a lane | odd column][lane + 2].
    __shared__ __attribute__((aligned(16))) f32x4 slab[4][2][2][2][68];

    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share an L2) and, within an XCD, start in the
    // order of blockIdx.  The strip plan (scale_2x_strip_plan) leans on both, for speed only: XCD x works on one
    // contiguous band of rows (a strip re-reads the last five input rows of the strip above it: L2 hits), and the
    // strips that start first are the longest.
    const int groupsPerRow = (colGroups + 3) >> 2;
    const int j = (int)(blockIdx.x >> 3);
    const int sj = j / groupsPerRow;                       // strip within the XCD's band, in dispatch order
    const int cg = (j - sj * groupsPerRow) * 4 + wv;
    int rBegin, n;                                         // emitting steps r = rBegin .. rBegin+n-1 (out of 2 .. inH+2)
    scale_2x_strip_of(inH, (int)(blockIdx.x & 7u), sj, stripsPerXcd, rBegin, n);   // scalar arithmetic, no table to wait for
    if (n <= 0 || cg >= colGroups) return;                 // wave-uniform; there is no workgroup barrier below
#ifdef LFG_DIAG_STAMPS
#define LFG_STAMP_ARGS , stamps, tStart, clk0, (int)(blockIdx.x & 7u) * stripsPerXcd + sj, wv
#else
#define LFG_STAMP_ARGS
#endif
    // Three straight-line bodies, one per strip length of the plan; a strip cut short at the end of its band runs the
    // next longer body with the surplus steps' stores switched off.
    if (n > LFG_SCALE_L1)
        scale_2x_strip<LFG_SCALE_L0, FUSED>(fs, in, inW, inH, inPitch, out, outW, outH, outPitch, classX, paletteX, weightY, rBegin, n, cg, lane, slab[wv] LFG_STAMP_ARGS);
    else if (n > LFG_SCALE_L2)
        scale_2x_strip<LFG_SCALE_L1, FUSED>(fs, in, inW, inH, inPitch, out, outW, outH, outPitch, classX, paletteX, weightY, rBegin, n, cg, lane, slab[wv] LFG_STAMP_ARGS);
    else
        scale_2x_strip<LFG_SCALE_L2, FUSED>(fs, in, inW, inH, inPitch, out, outW, outH, outPitch, classX, paletteX, weightY, rBegin, n, cg, lane, slab[wv] LFG_STAMP_ARGS);
#undef LFG_STAMP_ARGS
}

#ifdef LFG_DIAG_STAMPS
// Diagnostic build only (-DLFG_DIAG_STAMPS, never the shipped library): a device buffer for the per-wave time
// stamps of the LAST launch, dumped to a file by lfg_diag_dump_stamps().
static unsigned long long *g_stamps = nullptr;
static size_t g_stampWords = 0, g_stampWaves = 0;
static unsigned long long *diag_stamp_buffer(size_t waves, size_t perWave) {
    if (g_stampWords < waves * perWave) {
        if (g_stamps) (void)hipFree(g_stamps);
        (void)hipMalloc((void **)&g_stamps, waves * perWave * sizeof(unsigned long long));
        g_stampWords = waves * perWave;
    }
    (void)hipMemset(g_stamps, 0, waves * perWave * sizeof(unsigned long long));
    g_stampWaves = waves;
    return g_stamps;
}
extern "C" __attribute__((visibility("default"))) int lfg_diag_dump_stamps(const char *path) {
    if (!g_stamps) return -1;
    constexpr int per = LFG_SCALE_STEPS + 4;
    std::vector<unsigned long long> h(g_stampWaves * per);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "w");
    if (!f) return -2;
    for (size_t i = 0; i < g_stampWaves; ++i) {
        if (!h[per * i]) continue;                        // wave exited before stamping (surplus workgroup)
        fprintf(f, "%zu", i);
        for (int j = 0; j < per; ++j) fprintf(f, " %llu", h[per * i + j]);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}
#endif

bool scale_2x_supported(const lfg_frame &in, const lfg_frame &out) {
    return in.width % 2u == 0 && out.pitch % 16u == 0 &&
           (uintptr_t)out.data % 16u == 0 && in.pitch % 4u == 0 && (uintptr_t)in.data % 4u == 0 &&
           (uint64_t)in.height * in.pitch < 0x7fffffffull && (uint64_t)out.height * out.pitch < 0x7fffffffull;
}

int scale_2x_strips_per_xcd(int inH) {
    const int total = inH + 1, sum = LFG_SCALE_L0 + LFG_SCALE_L1 + LFG_SCALE_L2;
    int perXcd = 0;
    for (int x = 0; x < 8; ++x) {
        const int band = (total * (x + 1)) / 8 - (total * x) / 8;
        perXcd = std::max(perXcd, 3 * ((band + sum - 1) / sum));
    }
    return perXcd;
}

void scale_2x_strip_host(int inH, int xcd, int index, int &first, int &steps) {
    scale_2x_strip_of(inH, xcd, index, scale_2x_strips_per_xcd(inH), first, steps);
}

template <bool FUSED>
static hipError_t launch_2x(hipStream_t s, const FusedSource &fs, const lfg_frame &in, const lfg_frame &out,
                            const AxisTable &tx, const AxisTable &ty) {
    const int colGroups = ((int)in.width + kOwnedCols - 1) / kOwnedCols;
    const int groupsPerRow = (colGroups + 3) / 4;         // workgroups of four waves (four adjacent column groups)
    dim3 grid(8 * ty.strips_per_xcd * groupsPerRow);
    hipLaunchKernelGGL(scale_2x_kernel<FUSED>, grid, dim3(256), 0, s, fs,
                       (const uint8_t *)in.data, (int)in.width, (int)in.height, (int)in.pitch,
                       (uint8_t *)out.data, (int)out.width, (int)out.height, (int)out.pitch,
                       tx.d_class, tx.d_palette, ty.d_weight, colGroups, ty.strips_per_xcd
#ifdef LFG_DIAG_STAMPS
                       , diag_stamp_buffer((size_t)grid.x * 4, LFG_SCALE_STEPS + 4)
#endif
                       );
    return hipGetLastError();
}

hipError_t launch_scale_2x(hipStream_t s, const lfg_frame &in, const lfg_frame &out,
                           const AxisTable &tx, const AxisTable &ty) {
    return launch_2x<false>(s, FusedSource{}, in, out, tx, ty);
}

// interpolate(prev, curr, mv, t) upscaled 2x straight into `out` (all three inputs at input resolution).
hipError_t launch_interpolate_scale_2x(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                                       const lfg_frame &out, const AxisTable &tx, const AxisTable &ty, float factor, bool intended) {
    FusedSource fs{(const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                   (const int8_t *)mv.data, (int)mv.pitch, factor, intended ? 1 : 0};
    lfg_frame shape = curr;               // sizes only: the kernel reads no input frame
    return launch_2x<true>(s, fs, shape, out, tx, ty);
}

}  // namespace lfg
