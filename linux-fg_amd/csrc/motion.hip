// motion.hip -- per-pixel full-search block matching, the MI355X-native replacement of
// shaders/motion.comp (reference: /root/reference/shaders/motion.comp:16-57, dispatched by
// src/frame_manager.cpp:325-344 with blockSize = 8, searchRadius = 16.0f).
//
// For every pixel p and every candidate m in [-R,R]^2 (dy outer, dx inner, scan starts at (-R,-R)):
//     cost(p,m) = sum over the BxB block c = p - B/2 + (x,y), row-major, c inside the image, of
//                 distance(curr(c), prev(c+m)),        prev out of bounds -> (0,0,0,0)
// and the FIRST candidate reaching the minimum wins (strict '<', motion.comp:49).  The result must be
// bit-exact, and the fp32 sum is a single sequential chain, so no re-association is allowed: the
// tiled kernel keeps the literal summation order and only shares the per-pixel distances
//     d_m(c) = distance(curr(c), prev(c+m))
// between the up to B*B pixels whose blocks contain c.  Skipped (out-of-image) positions are added
// as +0.0f, which leaves a non-negative fp32 partial sum unchanged, so the chain is uniform.
//
// Roofline: fp32 VALU, not HBM: 64 dependent-order adds per (pixel, candidate) plus one
// correctly rounded distance per (position, candidate); HBM traffic is 10 B/pixel.
//
// motion_tiled_8_16_kernel (B = 8, R = 16): a 256-thread workgroup owns a 64x32 pixel tile.
//   LDS: prev tile with halo as packed RGBA8 (103 x 71 px), and a double-buffered plane
//   D[39][72] of distances for ONE candidate over the tile plus its block halo.
//   Per candidate k:   phase A  every thread computes ~11 entries of D_{k+1} (its curr texels stay in
//                               registers as floats for the whole kernel)
//                      phase B  every thread owns 8 horizontally adjacent pixels of one row and runs
//                               their 8 chains over D_k: 8 rows x 15 floats from LDS, 512 adds
//                      one __syncthreads().
// motion_generic_kernel: any block size / whole-number radius, one thread per pixel, literal loops
//   straight from global memory (slow; used for non-default parameters and as an on-device cross-check).
#include "lfg_device.hpp"
#include "lfg_internal.hpp"

namespace lfg {

// distance() of two texels given as packed RGBA8, oracle choices (1) and (7):
// sqrt(((dx*dx + dy*dy) + dz*dz) + dw*dw) with dx = a.x/255 - b.x/255, correctly rounded sqrt.
__device__ __forceinline__ float dist_f(float cx, float cy, float cz, float cw, uint32_t p) {
    const float dx = cx - unorm8_to_float(byte0(p));
    const float dy = cy - unorm8_to_float(byte1(p));
    const float dz = cz - unorm8_to_float(byte2(p));
    const float dw = cw - unorm8_to_float(byte3(p));
    return __builtin_sqrtf(((dx * dx + dy * dy) + dz * dz) + dw * dw);
}

// ------------------------------------------------------------------------------ tiled, B = 8, R = 16

constexpr int kB = 8, kR = 16;
constexpr int kTW = 64, kTH = 32;                 // pixel tile
constexpr int kDW = kTW + kB - 1;                 // 71 block positions across
constexpr int kDH = kTH + kB - 1;                 // 39 down
constexpr int kDP = 72;                           // D row pitch (floats)
constexpr int kPW = kDW + 2 * kR;                 // 103 prev columns
constexpr int kPH = kDH + 2 * kR;                 // 71 prev rows
constexpr int kPP = 104;                          // prev row pitch (pixels)
constexpr int kPos = (kDP * kDH + 255) / 256;     // D entries per thread (11)
constexpr int kCand = (2 * kR + 1) * (2 * kR + 1);

__global__ __launch_bounds__(256) void motion_tiled_8_16_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H) {
    __shared__ uint32_t sPrev[kPH * kPP];                              // 29.5 KB
    __shared__ __attribute__((aligned(16))) float sD[2][kDH * kDP];    // 22.5 KB

    const int tid = threadIdx.x;
    const int tx0 = blockIdx.x * kTW, ty0 = blockIdx.y * kTH;         // tile origin (pixels)
    const int bx0 = tx0 - kB / 2, by0 = ty0 - kB / 2;                 // image coords of D(0,0)

    // prev tile: LDS (row j, col i) = prev(bx0 - R + i, by0 - R + j), zero outside the image (choice 5).
    for (int i = tid; i < kPH * kPP; i += 256) {
        const int j = i / kPP, c = i - j * kPP;
        const int gx = bx0 - kR + c, gy = by0 - kR + j;
        uint32_t v = 0u;
        if (c < kPW && gx >= 0 && gy >= 0 && gx < W && gy < H)
            v = *reinterpret_cast<const uint32_t *>(prev + (size_t)gy * (size_t)prevPitch + (size_t)gx * 4u);
        sPrev[i] = v;
    }

    // This thread's D entries: linear index e = tid + 256 n  ->  (cy, cx) = (e / 72, e % 72).
    float cf[kPos][4];
    int pbase[kPos];             // LDS index of prev(c + (-R,-R)) for the entry's position c
    uint32_t valid = 0u;         // bit n: entry n is a real position inside the image
#pragma unroll
    for (int n = 0; n < kPos; ++n) {
        const int e = tid + 256 * n;
        const int cy = e / kDP, cx = e - cy * kDP;
        const int gx = bx0 + cx, gy = by0 + cy;
        const bool ok = (e < kDP * kDH) && (cx < kDW) && gx >= 0 && gy >= 0 && gx < W && gy < H;
        uint32_t p = 0u;
        if (ok) p = *reinterpret_cast<const uint32_t *>(curr + (size_t)gy * (size_t)currPitch + (size_t)gx * 4u);
        cf[n][0] = unorm8_to_float(byte0(p)); cf[n][1] = unorm8_to_float(byte1(p));
        cf[n][2] = unorm8_to_float(byte2(p)); cf[n][3] = unorm8_to_float(byte3(p));
        pbase[n] = ok ? (cy * kPP + cx) : 0;
        if (ok) valid |= (1u << n);
    }
    __syncthreads();

    auto phaseA = [&](int cand, float *__restrict__ D) {
        const int dyi = cand / (2 * kR + 1), dxi = cand - dyi * (2 * kR + 1);
        const int off = dyi * kPP + dxi;          // (dy + R) rows, (dx + R) columns
#pragma unroll
        for (int n = 0; n < kPos; ++n) {
            const int e = tid + 256 * n;
            if (e < kDP * kDH) {
                const uint32_t p = sPrev[pbase[n] + off];
                const float d = dist_f(cf[n][0], cf[n][1], cf[n][2], cf[n][3], p);
                D[e] = ((valid >> n) & 1u) ? d : 0.0f;
            }
        }
    };

    // Phase-B ownership: 8 pixels (tile row `ry`, columns 8*rxq .. 8*rxq+7).
    const int rxq = tid & 7, ry = tid >> 3;
    float best[8];
    int bestCand[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { best[i] = 1e10f; bestCand[i] = 0; }   // motion.comp:23-24

    phaseA(0, sD[0]);
    __syncthreads();

    for (int cand = 0; cand < kCand; ++cand) {
        if (cand + 1 < kCand) phaseA(cand + 1, sD[(cand + 1) & 1]);

        const float *__restrict__ D = sD[cand & 1];
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int y = 0; y < kB; ++y) {
            const float4 *row = reinterpret_cast<const float4 *>(D + (ry + y) * kDP + rxq * 8);
            const float4 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
            const float e[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w,
                                 q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
            for (int x = 0; x < kB; ++x) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += e[i + x];       // pixel i, block column x
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (acc[i] < best[i]) { best[i] = acc[i]; bestCand[i] = cand; }   // motion.comp:49-52
        }
        __syncthreads();
    }

    const int py = ty0 + ry, px0 = tx0 + rxq * 8;
    if (py < H && px0 < W) {
        int8_t o[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int dyi = bestCand[i] / (2 * kR + 1), dxi = bestCand[i] - dyi * (2 * kR + 1);
            o[2 * i] = (int8_t)(dxi - kR);
            o[2 * i + 1] = (int8_t)(dyi - kR);
        }
        int8_t *dst = mv + (size_t)py * (size_t)mvPitch + (size_t)px0 * 2u;
        if (px0 + 7 < W && (mvPitch & 15) == 0) {
            uint4 v;
            v.x = (uint8_t)o[0] | ((uint32_t)(uint8_t)o[1] << 8) | ((uint32_t)(uint8_t)o[2] << 16) | ((uint32_t)(uint8_t)o[3] << 24);
            v.y = (uint8_t)o[4] | ((uint32_t)(uint8_t)o[5] << 8) | ((uint32_t)(uint8_t)o[6] << 16) | ((uint32_t)(uint8_t)o[7] << 24);
            v.z = (uint8_t)o[8] | ((uint32_t)(uint8_t)o[9] << 8) | ((uint32_t)(uint8_t)o[10] << 16) | ((uint32_t)(uint8_t)o[11] << 24);
            v.w = (uint8_t)o[12] | ((uint32_t)(uint8_t)o[13] << 8) | ((uint32_t)(uint8_t)o[14] << 16) | ((uint32_t)(uint8_t)o[15] << 24);
            *reinterpret_cast<uint4 *>(dst) = v;
        } else {
            for (int i = 0; i < 8 && px0 + i < W; ++i) { dst[2 * i] = o[2 * i]; dst[2 * i + 1] = o[2 * i + 1]; }
        }
    }
}

hipError_t launch_motion_tiled_8_16(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                    const lfg_frame &mv) {
    dim3 grid((curr.width + kTW - 1) / kTW, (curr.height + kTH - 1) / kTH);
    hipLaunchKernelGGL(motion_tiled_8_16_kernel, grid, dim3(256), 0, s,
                       (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                       (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------ generic (literal)

__global__ __launch_bounds__(256) void motion_generic_kernel(
    const uint8_t *__restrict__ prev, int prevPitch, const uint8_t *__restrict__ curr, int currPitch,
    int8_t *__restrict__ mv, int mvPitch, int W, int H, int B, int R) {
    const int px = blockIdx.x * 64 + (threadIdx.x & 63);
    const int py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    const int bsx = px - B / 2, bsy = py - B / 2;
    float minDiff = 1e10f;
    int bx = 0, by = 0;
    for (int dy = -R; dy <= R; ++dy) {
        for (int dx = -R; dx <= R; ++dx) {
            float diff = 0.0f;
            for (int y = 0; y < B; ++y) {
                const int cy = bsy + y;
                if (cy < 0 || cy >= H) continue;
                for (int x = 0; x < B; ++x) {
                    const int cx = bsx + x;
                    if (cx < 0 || cx >= W) continue;
                    const uint32_t c = *reinterpret_cast<const uint32_t *>(curr + (size_t)cy * (size_t)currPitch + (size_t)cx * 4u);
                    const int qx = cx + dx, qy = cy + dy;
                    uint32_t p = 0u;
                    if (qx >= 0 && qy >= 0 && qx < W && qy < H)
                        p = *reinterpret_cast<const uint32_t *>(prev + (size_t)qy * (size_t)prevPitch + (size_t)qx * 4u);
                    diff += dist_f(unorm8_to_float(byte0(c)), unorm8_to_float(byte1(c)),
                                   unorm8_to_float(byte2(c)), unorm8_to_float(byte3(c)), p);
                }
            }
            if (diff < minDiff) { minDiff = diff; bx = dx; by = dy; }
        }
    }
    int8_t *dst = mv + (size_t)py * (size_t)mvPitch + (size_t)px * 2u;
    dst[0] = (int8_t)bx; dst[1] = (int8_t)by;
}

hipError_t launch_motion_generic(hipStream_t s, const lfg_frame &prev, const lfg_frame &curr,
                                 const lfg_frame &mv, int block_size, int radius) {
    dim3 grid((curr.width + 63) / 64, (curr.height + 3) / 4);
    hipLaunchKernelGGL(motion_generic_kernel, grid, dim3(256), 0, s,
                       (const uint8_t *)prev.data, (int)prev.pitch, (const uint8_t *)curr.data, (int)curr.pitch,
                       (int8_t *)mv.data, (int)mv.pitch, (int)curr.width, (int)curr.height, block_size, radius);
    return hipGetLastError();
}

}  // namespace lfg
