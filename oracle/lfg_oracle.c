/*
 * lfg_oracle.c -- CPU restatement of linux-fg's three compute shaders.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker.  The product path (linux-fg_amd/) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or known-answer
 * vectors for this path and cannot be built or run in this
 * image (no Vulkan loader/ICD, no glslc).  This file is therefore
 * pinned only by the analytic known-answer cases of SURVEY.md section 8(c)
 * (tests/test_oracle_kat.py) and by an independent numpy restatement on tiny
 * inputs (oracle/numpy_restatement.py).
 *
 * What is restated (all citations relative to /root/reference):
 *   lfg_oracle_scale        shaders/scale.comp:14-61        (Lanczos-3, 6x6 taps)
 *   lfg_oracle_motion       shaders/motion.comp:16-57       (full-search block match)
 *   lfg_oracle_interpolate  shaders/interpolate.comp:15-40  (MV-displaced blend)
 * with the parameters the host pushes: src/scaler.cpp:348-357,
 * src/frame_manager.cpp:325-334 (blockSize 8, searchRadius 16.0f) and :351-358.
 *
 * Driver-dependent behaviour that the GLSL text leaves open is fixed here, in
 * writing:
 *   (1) UNORM8 texel -> float  = (float)byte / 255.0f
 *   (2) float -> UNORM8 store  = clamp to [0,1], * 255.0f, round half to even
 *   (3) LINEAR / CLAMP_TO_EDGE sampler = Vulkan-spec bilinear with fp32 weights:
 *       u = s*W - 0.5, i0 = floor(u), a = u - i0, i1 = i0 + 1, both clamped to
 *       [0, W-1]; tau = (1-a)(1-b) t00 + a(1-b) t10 + (1-a)b t01 + ab t11
 *   (4) sin  = (float)sin((double)x)  (correctly rounded for all practical purposes)
 *       sqrt = IEEE correctly rounded sqrtf; no FMA contraction anywhere
 *       (compile with -ffp-contract=off, never -ffast-math)
 *   (5) texelFetch out of bounds (only `previousFrame` in motion.comp:43-44 can be)
 *       returns (0,0,0,0)
 *   (6) every fp32 sum is evaluated in source order, left to right
 *   (7) distance(a,b) = sqrt(((dx*dx + dy*dy) + dz*dz) + dw*dw), d = a - b
 *   (8) texture(motionVectors, uv) at an exact texel centre returns that texel
 *       (interpolate.comp:31)
 *   (9) mix(x, y, a) = x*(1-a) + y*a   (GLSL 4.50 section 8.3)
 *
 * Frames are tightly packed RGBA8, row-major, 4 bytes per pixel.  Motion vectors are
 * two floats per pixel (x, y) -- the .xy of the rgba32f image motion.comp:7 declares.
 * Every entry point takes a region of interest [x0,x1) x [y0,y1) in OUTPUT pixel
 * coordinates and writes only those pixels, so full-size spot checks stay cheap.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LFG_EXPORT __attribute__((visibility("default")))

typedef struct { float x, y, z, w; } vec4;

static inline float texel_to_float(uint8_t b) { return (float)b / 255.0f; }   /* choice (1) */

static inline uint8_t float_to_unorm8(float v) {                              /* choice (2) */
    if (!(v > 0.0f)) v = 0.0f;          /* also maps NaN to 0, as the Vulkan conversion rules do */
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)lrintf(v * 255.0f); /* default rounding mode = round half to even */
}

static inline vec4 fetch(const uint8_t *img, int W, int x, int y) {
    const uint8_t *p = img + ((size_t)y * (size_t)W + (size_t)x) * 4u;
    vec4 r = { texel_to_float(p[0]), texel_to_float(p[1]), texel_to_float(p[2]), texel_to_float(p[3]) };
    return r;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* choice (3): texture() on a LINEAR / CLAMP_TO_EDGE / normalized-coordinate sampler
 * (src/scaler.cpp:222-228, src/frame_manager.cpp:555-561). */
static vec4 texture_bilinear(const uint8_t *img, int W, int H, float s, float t) {
    float u = s * (float)W - 0.5f;
    float v = t * (float)H - 0.5f;
    float fu = floorf(u), fv = floorf(v);
    float a = u - fu, b = v - fv;
    int i0 = (int)fu, j0 = (int)fv;
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = clampi(i0, 0, W - 1); i1 = clampi(i1, 0, W - 1);
    j0 = clampi(j0, 0, H - 1); j1 = clampi(j1, 0, H - 1);
    vec4 t00 = fetch(img, W, i0, j0), t10 = fetch(img, W, i1, j0);
    vec4 t01 = fetch(img, W, i0, j1), t11 = fetch(img, W, i1, j1);
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b);
    float w01 = (1.0f - a) * b,          w11 = a * b;
    vec4 r;
    r.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
    r.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
    r.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
    r.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
    return r;
}

/* ---------------------------------------------------------------- scale.comp */

#define LANCZOS_A 3.0f                                   /* scale.comp:14 */

static inline float oracle_sinf(float x) { return (float)sin((double)x); }    /* choice (4) */

static float lanczos(float x) {                          /* scale.comp:16-20 */
    if (x == 0.0f) return 1.0f;
    float px = 3.14159265359f * x;
    return LANCZOS_A * oracle_sinf(px) * oracle_sinf(px / LANCZOS_A) / (px * px);
}

static vec4 sample_lanczos(const uint8_t *in, int inW, int inH, float uvx, float uvy) {
    /* scale.comp:22-49 */
    float tsx = 1.0f / (float)inW, tsy = 1.0f / (float)inH;
    float ppx = uvx * (float)inW - 0.5f, ppy = uvy * (float)inH - 0.5f;
    float fx = ppx - floorf(ppx), fy = ppy - floorf(ppy);             /* fract */
    float sx = floorf(ppx) - (LANCZOS_A - 1.0f), sy = floorf(ppy) - (LANCZOS_A - 1.0f);
    vec4 color = { 0.0f, 0.0f, 0.0f, 0.0f };
    float total = 0.0f;
    for (float y = 0.0f; y < 2.0f * LANCZOS_A; y++) {
        for (float x = 0.0f; x < 2.0f * LANCZOS_A; x++) {
            float spx = (sx + x + 0.5f) * tsx, spy = (sy + y + 0.5f) * tsy;
            if (spx < 0.0f || spy < 0.0f || spx > 1.0f || spy > 1.0f) continue;
            float dx = x - fx - (LANCZOS_A - 1.0f), dy = y - fy - (LANCZOS_A - 1.0f);
            float weight = lanczos(dx) * lanczos(dy);
            vec4 t = texture_bilinear(in, inW, inH, spx, spy);
            color.x += t.x * weight; color.y += t.y * weight;
            color.z += t.z * weight; color.w += t.w * weight;
            total += weight;
        }
    }
    color.x /= total; color.y /= total; color.z /= total; color.w /= total;
    return color;
}

typedef struct {
    int kind;                    /* 0 scale, 1 motion, 2 interpolate */
    const uint8_t *a, *b;        /* scale: a = in; motion/interp: a = prev, b = curr */
    const float *mv_in;
    uint8_t *out8;
    float *mv_out;
    int inW, inH, W, H;          /* W,H = output / image size */
    int blockSize; float searchRadius, factor;
    int semantics;               /* 0 = the shaders as written; 1 = "intended" (opt-in, see lfg_oracle_*_ex) */
    int x0, y0, x1, y1;
    int row_next;                /* work counter, protected by mu */
    pthread_mutex_t mu;
} job_t;

static void scale_row(const job_t *j, int py) {
    for (int px = j->x0; px < j->x1; px++) {             /* scale.comp:51-61 */
        float uvx = ((float)px + 0.5f) / (float)j->W;
        float uvy = ((float)py + 0.5f) / (float)j->H;
        vec4 c = sample_lanczos(j->a, j->inW, j->inH, uvx, uvy);
        uint8_t *o = j->out8 + ((size_t)py * (size_t)j->W + (size_t)px) * 4u;
        o[0] = float_to_unorm8(c.x); o[1] = float_to_unorm8(c.y);
        o[2] = float_to_unorm8(c.z); o[3] = float_to_unorm8(c.w);
    }
}

/* --------------------------------------------------------------- motion.comp */

static inline float distance4(vec4 p, vec4 q) {          /* choice (7) */
    float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z, dw = p.w - q.w;
    return sqrtf(((dx * dx + dy * dy) + dz * dz) + dw * dw);
}

static void motion_row(const job_t *j, int py) {
    const int W = j->W, H = j->H, bs = j->blockSize;
    const float R = j->searchRadius;
    for (int px = j->x0; px < j->x1; px++) {             /* motion.comp:16-57 */
        int bsx = px - bs / 2, bsy = py - bs / 2;
        float bestx = 0.0f, besty = 0.0f;
        float minDiff = 1e10f, bestD2 = 3.0e38f;
        for (float dy = -R; dy <= R; dy += 1.0f) {
            for (float dx = -R; dx <= R; dx += 1.0f) {
                int mx = (int)dx, my = (int)dy;          /* ivec2(motion): truncation */
                float diff = 0.0f;
                for (int y = 0; y < bs; y++) {
                    for (int x = 0; x < bs; x++) {
                        int cx = bsx + x, cy = bsy + y;
                        if (cx < 0 || cy < 0 || cx >= W || cy >= H) continue;
                        vec4 cc = fetch(j->b, W, cx, cy);
                        int qx = cx + mx, qy = cy + my;
                        vec4 pc = { 0.0f, 0.0f, 0.0f, 0.0f };              /* choice (5) */
                        if (qx >= 0 && qy >= 0 && qx < W && qy < H) pc = fetch(j->a, W, qx, qy);
                        diff += distance4(cc, pc);
                    }
                }
                if (j->semantics == 0) {
                    if (diff < minDiff) { minDiff = diff; bestx = dx; besty = dy; }         /* motion.comp:49-52 */
                } else {
                    /* intended: among equal costs the shortest vector wins, then scan order */
                    float d2 = dx * dx + dy * dy;
                    if (diff < minDiff || (diff == minDiff && d2 < bestD2)) { minDiff = diff; bestD2 = d2; bestx = dx; besty = dy; }
                }
            }
        }
        float *o = j->mv_out + ((size_t)py * (size_t)W + (size_t)px) * 2u;
        o[0] = bestx; o[1] = besty;                      /* imageStore(vec4(best,0,1)).xy */
    }
}

/* ---------------------------------------------------------- interpolate.comp */

static vec4 sample_with_motion(const uint8_t *img, int W, int H, float uvx, float uvy,
                               float mx, float my, float scale) {   /* interpolate.comp:15-22 */
    float sx = uvx + mx * scale, sy = uvy + my * scale;
    vec4 z = { 0.0f, 0.0f, 0.0f, 0.0f };
    if (sx < 0.0f || sy < 0.0f || sx > 1.0f || sy > 1.0f) return z;
    return texture_bilinear(img, W, H, sx, sy);
}

static inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }   /* choice (9) */

static void interpolate_row(const job_t *j, int py) {
    const int W = j->W, H = j->H;
    const float t = j->factor;
    for (int px = j->x0; px < j->x1; px++) {             /* interpolate.comp:24-40 */
        float uvx = ((float)px + 0.5f) / (float)W;
        float uvy = ((float)py + 0.5f) / (float)H;
        const float *m = j->mv_in + ((size_t)py * (size_t)W + (size_t)px) * 2u;  /* choice (8) */
        float mx = m[0], my = m[1];
        if (j->semantics != 0) { mx = mx / (float)W; my = my / (float)H; }       /* intended: pixels -> uv units */
        vec4 p = sample_with_motion(j->a, W, H, uvx, uvy, mx, my, -t);
        vec4 c = sample_with_motion(j->b, W, H, uvx, uvy, mx, my, 1.0f - t);
        uint8_t *o = j->out8 + ((size_t)py * (size_t)W + (size_t)px) * 4u;
        o[0] = float_to_unorm8(mixf(p.x, c.x, t)); o[1] = float_to_unorm8(mixf(p.y, c.y, t));
        o[2] = float_to_unorm8(mixf(p.z, c.z, t)); o[3] = float_to_unorm8(mixf(p.w, c.w, t));
    }
}

/* ------------------------------------------------------------- row scheduler */

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int py = j->row_next++;
        pthread_mutex_unlock(&j->mu);
        if (py >= j->y1) break;
        if (j->kind == 0) scale_row(j, py);
        else if (j->kind == 1) motion_row(j, py);
        else interpolate_row(j, py);
    }
    return NULL;
}

static int run(job_t *j, int nthreads) {
    if (j->x0 < 0 || j->y0 < 0 || j->x1 > j->W || j->y1 > j->H || j->x0 > j->x1 || j->y0 > j->y1) return -1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    j->row_next = j->y0;
    pthread_mutex_init(&j->mu, NULL);
    if (nthreads == 1) { worker(j); pthread_mutex_destroy(&j->mu); return 0; }
    pthread_t th[256];
    int started = 0;
    for (int i = 0; i < nthreads; i++) {
        if (pthread_create(&th[i], NULL, worker, j) != 0) break;
        started++;
    }
    if (started == 0) worker(j);
    for (int i = 0; i < started; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j->mu);
    return 0;
}

/* ----------------------------------------------------------------- exports */

LFG_EXPORT int lfg_oracle_scale(const uint8_t *in, int inW, int inH, uint8_t *out, int outW, int outH,
                                int x0, int y0, int x1, int y1, int nthreads) {
    if (!in || !out || inW <= 0 || inH <= 0 || outW <= 0 || outH <= 0) return -1;
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 0; j.a = in; j.out8 = out; j.inW = inW; j.inH = inH; j.W = outW; j.H = outH;
    j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1;
    return run(&j, nthreads);
}

/* semantics 0: motion.comp as written.  semantics 1 ("intended"; NOT the parity contract):
 * among candidates of equal cost the one with the smallest dx*dx + dy*dy wins, scan order breaking what is left,
 * so flat areas report (0,0) instead of (-R,-R) (F6). */
LFG_EXPORT int lfg_oracle_motion_ex(const uint8_t *prev, const uint8_t *curr, int W, int H,
                                    int blockSize, float searchRadius, float *mv_xy,
                                    int x0, int y0, int x1, int y1, int nthreads, int semantics) {
    if (!prev || !curr || !mv_xy || W <= 0 || H <= 0 || blockSize < 0 || !(searchRadius >= 0.0f)) return -1;
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 1; j.a = prev; j.b = curr; j.mv_out = mv_xy; j.W = W; j.H = H;
    j.blockSize = blockSize; j.searchRadius = searchRadius; j.semantics = semantics;
    j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1;
    return run(&j, nthreads);
}

LFG_EXPORT int lfg_oracle_motion(const uint8_t *prev, const uint8_t *curr, int W, int H,
                                 int blockSize, float searchRadius, float *mv_xy,
                                 int x0, int y0, int x1, int y1, int nthreads) {
    return lfg_oracle_motion_ex(prev, curr, W, H, blockSize, searchRadius, mv_xy, x0, y0, x1, y1, nthreads, 0);
}

/* semantics 0: interpolate.comp as written (pixel-unit motion added to normalised uv, F5).  semantics 1
 * ("intended"): the motion vector is divided by the image size first, so it displaces by pixels. */
LFG_EXPORT int lfg_oracle_interpolate_ex(const uint8_t *prev, const uint8_t *curr, const float *mv_xy,
                                         int W, int H, float factor, uint8_t *out,
                                         int x0, int y0, int x1, int y1, int nthreads, int semantics) {
    if (!prev || !curr || !mv_xy || !out || W <= 0 || H <= 0) return -1;
    job_t j; memset(&j, 0, sizeof j);
    j.kind = 2; j.a = prev; j.b = curr; j.mv_in = mv_xy; j.out8 = out; j.W = W; j.H = H;
    j.factor = factor; j.semantics = semantics;
    j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1;
    return run(&j, nthreads);
}

LFG_EXPORT int lfg_oracle_interpolate(const uint8_t *prev, const uint8_t *curr, const float *mv_xy,
                                      int W, int H, float factor, uint8_t *out,
                                      int x0, int y0, int x1, int y1, int nthreads) {
    return lfg_oracle_interpolate_ex(prev, curr, mv_xy, W, H, factor, out, x0, y0, x1, y1, nthreads, 0);
}

/* Per-axis raw Lanczos weights for one output coordinate, as scale.comp:24-41 computes
 * them: returns start index (floor(pixelPos) - 2) and six weights L(k - f - 2).  Used by
 * the known-answer tests (S-KAT3: phase weights at exact 2x). */
LFG_EXPORT int lfg_oracle_lanczos_taps(int p, int inSize, int outSize, float weights6[6]) {
    float uv = ((float)p + 0.5f) / (float)outSize;
    float pp = uv * (float)inSize - 0.5f;
    float f = pp - floorf(pp);
    float s = floorf(pp) - (LANCZOS_A - 1.0f);
    for (int k = 0; k < 6; k++) weights6[k] = lanczos((float)k - f - (LANCZOS_A - 1.0f));
    return (int)s;
}
