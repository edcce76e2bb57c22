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
// 0), so the thread spends nothing on coordinates, range tests or weights: eight loads, mix and the packing -- the same two
// functions as everywhere else.  (tools/bench_blend.hip: the shape's floor on this chip, and the load forms against each other.)
// The columns whose uv misses its texel centre by an ulp (102 of 3840, in 47 groups of four) are scattered over most waves of a
// row.  In a centre ROW such a pixel is a blend of two neighbouring texels of its own row (fraction b = 0: the second row's two
// products are exactly 0), so a lane whose quad holds one -- the host's mask, one scalar load per wave -- fetches the quad's
// two neighbours as well and runs its pixels through `row_blend` below: the bilinear sum of csrc/lfg_interp.hpp with the terms
// that are exactly zero left out.  (What did NOT work, each measured: those quads in extra blocks of their own -- 28 us of a 53 us
// launch, every one of their loads fetched a line some other XCD had fetched already; per-column tables read by those lanes
// -- the same few L2 lines asked for by every wave of the chip; a patch for ONE such pixel per quad -- they come in runs, two
// thirds of the quads hold several and fell back to the generic code.)
// Rows whose uv misses (85 of 2160) do the same with two texel rows (`cell_blend`), the whole wave together.
constexpr int kRsrcRgba8Unorm = (int)(4u | (5u << 3) | (6u << 6) | (7u << 9) | (0u << 12) | (10u << 15));     // DST_SEL RGBA, NUM_FORMAT UNORM, DATA_FORMAT 8_8_8_8

__device__ __forceinline__ uint32_t mix_pack(const V4 p, const V4 c, float t) {
    return pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
}

// A pixel of a centre row at zero displacement whose COLUMN misses its texel centre: texture() at (uvx, the row's uvy) is
// ((w00 t00 + w10 t10) + w01 t01) + w11 t11 with b = 0, i.e. w00 = (1 - a) * 1, w10 = a * 1, w01 = w11 = 0 and j0 = py: the sum
// is fl(fl(w00 t00) + fl(w10 t10)) -- adding the two exact zeros changes nothing (all terms are >= 0) -- of two neighbouring
// texels of the pixel's own row: u = uvx W - 0.5 lies within 0.001 of the pixel's column p for any width below a million, so
// floor(u) is p (a tiny) or p - 1 (a just below 1).  A column that does hit has a = 0 and comes out as t00 by the same
// arithmetic, so a lane that owns any such column runs all four of its pixels through it.
struct RowTaps { bool below; float w00, w10; };          // below: floor(u) = p - 1, the texels are p - 1 and p; else p and p + 1

// ((float)p + 0.5f) / (float)size, interpolate.comp:30: three instructions where the host has checked that they give the IEEE
// quotient for every p of the axis (lfg_internal.hpp: UvTable::rcp), the host's table otherwise.
__device__ __forceinline__ float uv_of(int p, int size, float rcp, int rcpExact, const float *__restrict__ table) {
#ifdef LFG_DIAG_UV_TABLE            // (A/B: always the table)
    return table[p];
#endif
    if (!rcpExact) return table[p];
    const float x = (float)p + 0.5f, q0 = x * rcp;
    return __builtin_fmaf(__builtin_fmaf(-q0, (float)size, x), rcp, q0);
}

// ... and by the division itself (the few lanes of the fast path that need it: measured 1.7 us faster there than the form above
// with its run-time switch, 20.7 against 22.4 us per 4K launch)
__device__ __forceinline__ float uv_div(int p, int size) { return ((float)p + 0.5f) / (float)size; }

__device__ __forceinline__ RowTaps row_taps(int p, int W, float uvx) {
    const float pk = (float)p;
    const float u = uvx * (float)W - 0.5f;
    const float fu = __builtin_floorf(u);
    const float a = u - fu;
    return RowTaps{fu < pk, (1.0f - a) * (1.0f - 0.0f), a * (1.0f - 0.0f)};
}

__device__ __forceinline__ V4 row_blend(const RowTaps &t, const f32x4 left, const f32x4 mid, const f32x4 right) {
    const f32x4 t00 = t.below ? left : mid, t10 = t.below ? mid : right;
    return V4{t.w00 * t00.x + t.w10 * t10.x, t.w00 * t00.y + t.w10 * t10.y, t.w00 * t00.z + t.w10 * t10.z, t.w00 * t00.w + t.w10 * t10.w};
}

// The same for a ROW that misses (85 of 2160): two texel rows, weights (1 - b) and b, and in a column that misses as well all
// four products -- texture_bilinear's sum with its weights formed from the same factors: w00 = (1 - a)(1 - b), w10 = a (1 - b),
// w01 = (1 - a) b, w11 = a b.  In a column that hits a = 0: w10 and w11 are exactly 0 and their terms add nothing.
struct QuadRows { uint32_t t[2][6]; };     // [row][left neighbour, the quad's four texels, right neighbour]

__device__ __forceinline__ V4 cell_blend(const RowTaps &tx, const RowTaps &ty, const QuadRows &q, int i) {
    const float w00 = tx.w00 * ty.w00, w10 = tx.w10 * ty.w00, w01 = tx.w00 * ty.w10, w11 = tx.w10 * ty.w10;
    const V4 t00 = unorm4(tx.below ? q.t[0][i] : q.t[0][i + 1]), t10 = unorm4(tx.below ? q.t[0][i + 1] : q.t[0][i + 2]);
    const V4 t01 = unorm4(tx.below ? q.t[1][i] : q.t[1][i + 1]), t11 = unorm4(tx.below ? q.t[1][i + 1] : q.t[1][i + 2]);
    V4 r;
    r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
    r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
    r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
    r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
    return r;
}

__device__ __forceinline__ QuadRows load_quad_rows(const uint8_t *__restrict__ img, int pitch, int W, int r0, int r1, int px0, bool edges) {
    const uint8_t *a = img + (size_t)r0 * (size_t)pitch, *b = img + (size_t)r1 * (size_t)pitch;
    const uint4 qa = *reinterpret_cast<const uint4 *>(a + (size_t)px0 * 4u), qb = *reinterpret_cast<const uint4 *>(b + (size_t)px0 * 4u);
    QuadRows q;
    q.t[0][0] = q.t[0][5] = q.t[1][0] = q.t[1][5] = 0u;       // (not fetched for a quad of centre columns: their weights are exactly 0 there)
    if (edges) {
        const size_t xl = (size_t)max(px0 - 1, 0) * 4u, xr = (size_t)min(px0 + 4, W - 1) * 4u;
        q.t[0][0] = *reinterpret_cast<const uint32_t *>(a + xl); q.t[0][5] = *reinterpret_cast<const uint32_t *>(a + xr);
        q.t[1][0] = *reinterpret_cast<const uint32_t *>(b + xl); q.t[1][5] = *reinterpret_cast<const uint32_t *>(b + xr);
    }
    q.t[0][1] = qa.x; q.t[0][2] = qa.y; q.t[0][3] = qa.z; q.t[0][4] = qa.w;
    q.t[1][1] = qb.x; q.t[1][2] = qb.y; q.t[1][3] = qb.z; q.t[1][4] = qb.w;
    return q;
}

// The generic path, one pixel at a time: both samples' addresses and weights first, then their eight texel loads, then the
// arithmetic (csrc/lfg_interp.hpp: sample_taps / sample_load / sample_finish -- interpolate.comp:15-22, 34-38 in two steps; round
// 3's per-sample branches put up to four memory round trips per pixel in a row).  Pixel by pixel rather than all 32 loads of a
// quad at once: that took 126 registers and halved the occupancy of the fast path beside it.  A pixel neither of whose samples
// lies inside the image -- the literal semantics under any motion, F5 -- loads nothing and is (0,0,0,0).
template <bool INTENDED>
__device__ __forceinline__ bool pixel_taps(const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
                                           int W, int H, int8_t mxi, int8_t myi, float uvx, float uvy, float t, SampleTaps &sp, SampleTaps &sc) {
    float mx = (float)mxi, my = (float)myi;
    if (INTENDED) { mx = mx / (float)W; my = my / (float)H; }
    // (the range tests first, on their own: under the literal semantics any motion puts both samples outside, and the pixel is
    //  done after a dozen instructions)
    if (!sample_inside(uvx, uvy, mx, my, -t) && !sample_inside(uvx, uvy, mx, my, 1.0f - t)) return false;
    sp = sample_taps(prev, W, H, prevPitch, uvx, uvy, mx, my, -t);
    sc = sample_taps(curr, W, H, currPitch, uvx, uvy, mx, my, 1.0f - t);
    return sp.inside || sc.inside;
}

__device__ __forceinline__ uint32_t pixel_finish(const SampleTaps &sp, const SampleTexels &tp, const SampleTaps &sc, const SampleTexels &tc, float t) {
    const V4 p = sample_finish(sp, tp);
    const V4 c = sample_finish(sc, tc);
    return pack_rgba8_unorm(mixf(p.x, c.x, t), mixf(p.y, c.y, t), mixf(p.z, c.z, t), mixf(p.w, c.w, t));
}

// tb.centreY[py] for a row that is the same for the whole wave: a scalar load of the aligned word around it.
__device__ __forceinline__ bool row_is_centre(const InterpTables &tb, int py) {
    const int row = __builtin_amdgcn_readfirstlane(py);
    const uint32_t w = reinterpret_cast<const uint32_t *>(tb.centreY)[row >> 2];
    return ((w >> (8 * (row & 3))) & 1u) != 0u;
}

// The host's verdict on whether the fast path may be used for these frames at all (launch_interpolate*): the frames below 2 GiB
// (32-bit buffer offsets); the vectors' base and pitch multiples of 8 (one 8-byte load per quad) and the output's of 16 (fast_output: the
// guarded 16-byte stores).  Of prev and curr only the BASE is asked to be 4-byte aligned and nothing of the pitch: their 16-byte
// loads (load_quad_rows) are global loads, which gfx950 serves at any 4-byte address -- a pitch of W * 4 + 4 is legal and tested
// (tests/test_gpu_parity.py: test_interpolate_fast_path_with_rows_that_are_only_4_byte_aligned); it costs bandwidth, not results.
static bool fast_frames(const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv) {
    const auto small = [](const lfg_frame &f) { return (uint64_t)f.pitch * f.height < (1ull << 31); };
    return small(prev) && small(curr) && (uintptr_t)prev.data % 4u == 0 && (uintptr_t)curr.data % 4u == 0 &&
           mv.pitch % 8u == 0 && (uintptr_t)mv.data % 8u == 0;
}
static bool fast_output(const lfg_frame &o) { return o.pitch % 16u == 0 && (uintptr_t)o.data % 16u == 0; }

#ifndef LFG_INTERP_WAVES
#define LFG_INTERP_WAVES 6            // waves per SIMD the register allocation aims for (hipcc: the second argument of __launch_bounds__)
#endif
#ifndef LFG_INTERP_WAVES_MULTI
#define LFG_INTERP_WAVES_MULTI 6      // ... with several factors (at 8 both kinds spill; 5 to 7 measured alike)
#endif
// INTENDED = false: interpolate.comp as written.  INTENDED = true (opt-in, lfg_set_semantics):
// the motion vector is divided by the image size before it is added to uv, so it displaces by pixels.
// N = 1: lfg_interpolate.  N > 1 (BASELINE config 5, 60 -> 240 fps: t = 1/4, 1/2, 3/4 from ONE motion field): one thread reads
// its four vectors once and produces the pixels of all N frames -- 10 + 4 N B/pixel instead of 14 N; where the vector is (0,0)
// the texels are fetched once and only mix() is repeated.  Every output byte is computed by the same functions, in the same
// order, whatever N: the frames are identical to N separate calls (tests/test_gpu_parity.py).
constexpr int kMaxMulti = 4;
struct MultiTargets {
    uint8_t *out[kMaxMulti];
    int pitch[kMaxMulti];
    float t[kMaxMulti];
};

template <int N, bool INTENDED>
__global__ __launch_bounds__(256, N == 1 ? LFG_INTERP_WAVES : LFG_INTERP_WAVES_MULTI) void interpolate_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    const int8_t *__restrict__ mv, int mvPitch, MultiTargets tg, int W, int H, bool wideStores,
    InterpTables tb, int fast) {
    const int lane = threadIdx.x & 63;
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (py >= H) return;
    const int px0 = (blockIdx.x * 64 + lane) * 4;              // a group of four pixels
    if (px0 >= W) return;
#ifdef LFG_DIAG_INTERP_ROWS_CENTRE      // (timing experiment, wrong pixels: every row a centre row)
    const bool rowCentre = fast;
#else
    const bool rowCentre = fast && row_is_centre(tb, py);
#endif
    const int8_t *mrow = mv + (size_t)py * (size_t)mvPitch;
    const bool full = (px0 + 3 < W) && ((mvPitch & 7) == 0) && wideStores;

    int8_t m[8];
    uint32_t o[N][4];
    bool zero = false;                    // four zero vectors: the sample positions do not depend on t
    if (full) {
        const uint2 mm = *reinterpret_cast<const uint2 *>(mrow + (size_t)px0 * 2u);
        zero = (mm.x | mm.y) == 0u;
        if (rowCentre && zero) {
            // the quad fast path: four texels of each frame -- as floats, byte / 255.0f, from the texture-address unit: format-
            // converting loads through an 8_8_8_8 UNORM descriptor (bit for bit the oracle's conversion,
            // test_interpolate_all_byte_values_round_trip; the unpacking was a fifth of the wave's VALU work) -- and N blends
            const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(prev), 0, H * prevPitch, kRsrcRgba8Unorm);
            const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(curr), 0, H * currPitch, kRsrcRgba8Unorm);
            const int offP = py * prevPitch + px0 * 4, offC = py * currPitch + px0 * 4;
            f32x4 tp[6], tc[6];                  // [left neighbour, the quad, right neighbour]
#pragma unroll
            for (int i = 0; i < 4; ++i) tp[i + 1] = buffer_load_rgba8_format(rP, offP + 4 * i, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) tc[i + 1] = buffer_load_rgba8_format(rC, offC + 4 * i, 0);
#ifdef LFG_DIAG_INTERP_NO_PATCH         // (timing experiment, wrong pixels: every column a centre column)
            const bool laneGood = true;
#else
            const bool laneGood = ((tb.goodMask[blockIdx.x] >> lane) & 1ull) != 0ull;            // (one scalar load per wave)
#endif
            V4 fp[4], fc[4];
            if (laneGood) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { fp[i] = V4{tp[i + 1].x, tp[i + 1].y, tp[i + 1].z, tp[i + 1].w}; fc[i] = V4{tc[i + 1].x, tc[i + 1].y, tc[i + 1].z, tc[i + 1].w}; }
            } else {
                // a quad with columns that miss (see row_taps): its two neighbours in the row as well (asked for before anything
                // waits for the quad's own texels), and every pixel of it as the blend it is
                const int xl = max(px0 - 1, 0) * 4, xr = min(px0 + 4, W - 1) * 4;
                tp[0] = buffer_load_rgba8_format(rP, py * prevPitch + xl, 0); tp[5] = buffer_load_rgba8_format(rP, py * prevPitch + xr, 0);
                tc[0] = buffer_load_rgba8_format(rC, py * currPitch + xl, 0); tc[5] = buffer_load_rgba8_format(rC, py * currPitch + xr, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const RowTaps t = row_taps(px0 + i, W, uv_div(px0 + i, W));
                    fp[i] = row_blend(t, tp[i], tp[i + 1], tp[i + 2]);
                    fc[i] = row_blend(t, tc[i], tc[i + 1], tc[i + 2]);
                }
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[k][i] = mix_pack(fp[i], fc[i], tg.t[k]);
                *reinterpret_cast<uint4 *>(tg.out[k] + (size_t)py * (size_t)tg.pitch[k] + (size_t)px0 * 4u) = uint4{o[k][0], o[k][1], o[k][2], o[k][3]};
            }
            return;
        }
        if (fast && zero) {
            // a row that misses (the whole wave): the same with two texel rows
            const bool laneGood = ((tb.goodMask[blockIdx.x] >> lane) & 1ull) != 0ull;
            const RowTaps ty = row_taps(py, H, uv_div(py, H));
            const int j0 = ty.below ? py - 1 : py;
            const int r0 = clampi(j0, 0, H - 1), r1 = clampi(j0 + 1, 0, H - 1);
            const QuadRows qp = load_quad_rows(prev, prevPitch, W, r0, r1, px0, !laneGood);
            const QuadRows qc = load_quad_rows(curr, currPitch, W, r0, r1, px0, !laneGood);
            V4 fp[4], fc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                RowTaps tx = RowTaps{false, 1.0f, 0.0f};
                if (!laneGood) tx = row_taps(px0 + i, W, uv_div(px0 + i, W));
                fp[i] = cell_blend(tx, ty, qp, i);
                fc[i] = cell_blend(tx, ty, qc, i);
            }
#pragma unroll
            for (int k = 0; k < N; ++k) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[k][i] = mix_pack(fp[i], fc[i], tg.t[k]);
                *reinterpret_cast<uint4 *>(tg.out[k] + (size_t)py * (size_t)tg.pitch[k] + (size_t)px0 * 4u) = uint4{o[k][0], o[k][1], o[k][2], o[k][3]};
            }
            return;
        }
        m[0] = (int8_t)(mm.x & 0xff); m[1] = (int8_t)((mm.x >> 8) & 0xff);
        m[2] = (int8_t)((mm.x >> 16) & 0xff); m[3] = (int8_t)(mm.x >> 24);
        m[4] = (int8_t)(mm.y & 0xff); m[5] = (int8_t)((mm.y >> 8) & 0xff);
        m[6] = (int8_t)((mm.y >> 16) & 0xff); m[7] = (int8_t)(mm.y >> 24);
#ifdef LFG_DIAG_MV_UNIFORM             // (timing experiment, DESIGN.md 4.4: what reading the vectors costs -- every pixel takes the
                                       //  vector of the frame's centre, one scalar load; right for the benchmark's pan but for its rim)
        {
            const int8_t *centre = mv + (size_t)(H / 2) * (size_t)mvPitch + (size_t)(W / 2) * 2u;
            const int8_t cx = centre[0], cy = centre[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) { m[2 * i] = cx; m[2 * i + 1] = cy; }
            zero = (cx | cy) == 0;
        }
#endif
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = min(px0 + i, W - 1);
            m[2 * i] = mrow[(size_t)px * 2u]; m[2 * i + 1] = mrow[(size_t)px * 2u + 1];
        }
    }

#ifndef LFG_INTERP_SURE_OUT
#define LFG_INTERP_SURE_OUT 1
#endif
    // Under the literal semantics (F5) a vector is added to the NORMALISED coordinate in pixels: a component with |m * scale| >= 1
    // puts the sample outside [0,1] wherever the pixel lies -- uv is at least 0.5 / size above 0 and below 1, far more than an
    // ulp of 1, and rounding is monotone: uv + d > 1 for d >= 1, uv + d < 0 for d <= -1 -- so the pixel is (0,0,0,0) without its
    // coordinate ever being computed.  A quad whose four pixels are like that for both samples and every factor stores its zeros
    // and is done: the benchmark's pan (3 texels of displacement at t = 0.5) from 17.5 us a 4K frame towards what reading the
    // vectors and writing the frame costs.  (The same products as sample_inside: m * scale in fp32, scale = -t and 1 - t.)
    if (LFG_INTERP_SURE_OUT && !INTENDED && !zero) {
        bool allOut = true;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float t = tg.t[k], sa = -t, sb = 1.0f - t;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float mx = (float)m[2 * i], my = (float)m[2 * i + 1];
                const bool outA = __builtin_fabsf(mx * sa) >= 1.0f || __builtin_fabsf(my * sa) >= 1.0f;
                const bool outB = __builtin_fabsf(mx * sb) >= 1.0f || __builtin_fabsf(my * sb) >= 1.0f;
                allOut = allOut && outA && outB;
            }
        }
        if (allOut) {
#pragma unroll
            for (int k = 0; k < N; ++k) {
                uint8_t *orow = tg.out[k] + (size_t)py * (size_t)tg.pitch[k];
                if (full) {
                    *reinterpret_cast<uint4 *>(orow + (size_t)px0 * 4u) = uint4{0u, 0u, 0u, 0u};
                } else {
                    for (int i = 0; i < 4 && px0 + i < W; ++i) *reinterpret_cast<uint32_t *>(orow + (size_t)(px0 + i) * 4u) = 0u;
                }
            }
            return;
        }
    }
    // ((float)p + 0.5f) / (float)size, interpolate.comp:30, from the host's tables (lfg_internal.hpp: UvTable; the tables are
    // padded to whole groups of four)
    const float uvy = uv_of(py, H, tb.rcpH, tb.rcpExact, tb.uvy);
    float uvxOf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) uvxOf[i] = uv_of(min(px0 + i, W - 1), W, tb.rcpW, tb.rcpExact, tb.uvx);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool still = m[2 * i] == 0 && m[2 * i + 1] == 0;       // uv + 0 * scale is uv whatever the factor: one set of texels for all N
        SampleTaps sp, sc;
        SampleTexels tp, tc;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const float t = tg.t[k];
            if (k == 0 || !still) {
                o[k][i] = 0u;
                if (!pixel_taps<INTENDED>(prev, prevPitch, curr, currPitch, W, H, m[2 * i], m[2 * i + 1], uvxOf[i], uvy, t, sp, sc)) continue;
                tp = sample_load(sp); tc = sample_load(sc);
            }
            o[k][i] = pixel_finish(sp, tp, sc, tc, t);
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
static void launch_n(hipStream_t s, dim3 grid, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                     const MultiTargets &tg, int W, int H, bool wide, bool intended, const InterpTables &tb, int fast) {
    if (intended)
        hipLaunchKernelGGL((interpolate_kernel<N, true>), grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, tg, W, H, wide, tb, fast);
    else
        hipLaunchKernelGGL((interpolate_kernel<N, false>), grid, dim3(256), 0, s,
                           (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                           (const int8_t *)mv.data, (int)mv.pitch, tg, W, H, wide, tb, fast);
}

// `count` frames from one pass per group of up to kMaxMulti factors.
hipError_t launch_interpolate_multi(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr, const lfg_frame &mv,
                                    const lfg_frame *const *outs, const float *factors, int count, bool intended,
                                    const InterpTables &tb) {
    const int W = (int)curr.width, H = (int)curr.height;
    const int quads = (W + 3) / 4;
    const dim3 grid((quads + 63) / 64, (H + 3) / 4);
    for (int first = 0; first < count; first += kMaxMulti) {
        const int n = count - first < kMaxMulti ? count - first : kMaxMulti;
        MultiTargets tg{};
        bool wide = true;
        for (int k = 0; k < kMaxMulti; ++k) {
            const lfg_frame &o = *outs[first + (k < n ? k : 0)];
            tg.out[k] = (uint8_t *)o.data; tg.pitch[k] = (int)o.pitch; tg.t[k] = factors[first + (k < n ? k : 0)];
            wide = wide && fast_output(o);
        }
        const int fast = fast_frames(prev, curr, mv) && wide && tb.goodMask && W < (1 << 20) && H < (1 << 20) ? 1 : 0;
        switch (n) {
            case 1: launch_n<1>(s, grid, prev, curr, mv, tg, W, H, wide, intended, tb, fast); break;
            case 2: launch_n<2>(s, grid, prev, curr, mv, tg, W, H, wide, intended, tb, fast); break;
            case 3: launch_n<3>(s, grid, prev, curr, mv, tg, W, H, wide, intended, tb, fast); break;
            default: launch_n<4>(s, grid, prev, curr, mv, tg, W, H, wide, intended, tb, fast); break;
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_interpolate(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                              const lfg_frame &mv, const lfg_frame &out, float factor, bool intended, const InterpTables &tb) {
    const lfg_frame *outs[1] = {&out};
    return launch_interpolate_multi(s, prev, curr, mv, outs, &factor, 1, intended, tb);
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
