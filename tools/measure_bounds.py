#!/usr/bin/env python3
"""On the GPU box: how many candidates would a lower bound on the block-match cost remove, on the benchmark's own upscaled
frames?  For a sample of interior 16 x 56 segments (the prefilter's wave) the script computes, on the host, every
candidate's cost for every pixel -- sum over the 8 x 8 block of the texel distances, as shaders/motion.comp:33-46 adds
them -- takes each pixel's FINAL minimum as its threshold (the most any bound can be tested against) and counts the
candidates that a bound clears

  * for a pixel on its own, and
  * for the whole wave at once (every pixel of the segment cleared: what lets a wave skip a full evaluation),

for   sea     the successive-elimination bound |sum ||c|| - sum ||p||| <= sum ||c - p|| on box sums of texel norms,
      one     one distance of the block (the prefilter's 14-point lattice: every block holds one point),
      four    the four distances of the 4 x 4 lattice inside the block,
      sixteen the sixteen distances of the 2 x 2 lattice inside the block,
      four_sad, sixteen_sad  the same with every distance replaced by half the sum of its absolute differences (<= the distance).

The frames come from lfg_scale on the synthetic 1080p content (the bench's), nothing else runs on the GPU.
usage: measure_bounds.py [content ...]        (default: noisy occluded objects)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib.util  # noqa: E402

import numpy as np  # noqa: E402

from linux_fg_amd import capi  # noqa: E402

_spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)

R, B, SEG_H, SEG_W = 16, 8, 16, 56
N_SEG = int(os.environ.get("LFG_BOUND_SEGMENTS", "24"))


def box(a, k=B):
    """sums over k x k windows starting at every position (valid part)"""
    c = np.cumsum(np.cumsum(np.pad(a, ((1, 0), (1, 0))), 0), 1)
    return c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]


def lattice(d, step, x0, y0):
    """for every pixel of the segment: the sum of the distances at the block positions = (x0, y0) mod step inside its block
    (d: distances at the segment's 23 x 63 block positions; a pixel's block starts at its own position)"""
    m = np.zeros_like(d)
    m[y0::step, x0::step] = d[y0::step, x0::step]
    return box(m)


def segment_stats(prev, curr, sx, sy, unmatched_only):
    c = curr[sy - 4:sy - 4 + SEG_H + B - 1, sx - 4:sx - 4 + SEG_W + B - 1].astype(np.float32)       # the block positions
    cn = np.sqrt((c * c).sum(-1))
    cbox = box(cn)
    costs = np.empty((2 * R + 1, 2 * R + 1, SEG_H, SEG_W), np.float32)
    bounds = {k: np.empty_like(costs) for k in ("sea", "one", "four", "four_sad", "sixteen", "sixteen_sad")}
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            p = prev[sy - 4 + dy:sy - 4 + dy + SEG_H + B - 1, sx - 4 + dx:sx - 4 + dx + SEG_W + B - 1].astype(np.float32)
            d = np.sqrt(((p - c) ** 2).sum(-1))
            costs[dy + R, dx + R] = box(d)
            bounds["sea"][dy + R, dx + R] = np.abs(cbox - box(np.sqrt((p * p).sum(-1))))
            bounds["one"][dy + R, dx + R] = lattice(d, 8, 7, 7)
            bounds["four"][dy + R, dx + R] = lattice(d, 4, 3, 3)
            bounds["sixteen"][dy + R, dx + R] = lattice(d, 2, 0, 0)
            half_sad = 0.5 * np.abs(p - c).sum(-1)               # a distance is at least half the sum of its absolute differences
            bounds["four_sad"][dy + R, dx + R] = lattice(half_sad, 4, 3, 3)
            bounds["sixteen_sad"][dy + R, dx + R] = lattice(half_sad, 2, 0, 0)
    thr = costs.reshape(-1, SEG_H, SEG_W).min(0)
    out = {"thr_median": float(np.median(thr)), "thr_max": float(thr.max()), "unmatched": float((thr >= 2048).mean())}
    for k, b in bounds.items():
        cleared = b > thr[None, None] * 1.0001
        out[k + "_pixel"] = float(cleared.mean())
        out[k + "_wave"] = float(cleared.all((2, 3)).mean())
        if unmatched_only and (thr >= 2048).any():
            out[k + "_pixel_unmatched"] = float(cleared[:, :, thr >= 2048].mean())
    return out


def main():
    contents = sys.argv[1:] or ["noisy", "occluded", "objects"]
    ctx = capi.Context(0)
    w, h = 1920, 1080
    for content in contents:
        prev_in, curr_in = _bench.make_content(content, w, h, 0, 0)
        P, C = ctx.create_frame(2 * w, 2 * h), ctx.create_frame(2 * w, 2 * h)
        ctx.scale(ctx.frame_from(prev_in), P)
        ctx.scale(ctx.frame_from(curr_in), C)
        ctx.sync()
        prev, curr = ctx.download(P), ctx.download(C)
        rng = np.random.default_rng(4242)
        rows = []
        if content in ("occluded", "objects"):
            # segments on the rim of the patches: where the pan's vector stops matching (the bench's own patch positions)
            prng = np.random.default_rng((20240 if content == "occluded" else 30240) + 0)
            spots = []
            for _ in range(24):
                pw, ph = int(prng.integers(w // 60, w // 12)), int(prng.integers(h // 60, h // 12))
                x0, y0 = int(prng.integers(40, w - 40 - pw)), int(prng.integers(40, h - 40 - ph))
                if content == "objects":
                    prng.integers(-7, 8), prng.integers(-7, 8)
                spots += [(2 * x0 - 20, 2 * y0 + ph), (2 * x0 + pw, 2 * y0 - 6), (2 * x0 + pw, 2 * y0 + ph)]      # left edge, top edge, inside
            picks = [spots[i] for i in rng.permutation(len(spots))[:N_SEG]]
        else:
            picks = [(int(rng.integers(100, 2 * w - 200)), int(rng.integers(100, 2 * h - 200))) for _ in range(N_SEG)]
        for sx, sy in picks:
            sx = min(max(sx // 56 * 56, 56), 2 * w - 2 * 56)
            sy = min(max(sy // 16 * 16, 32), 2 * h - 64)
            rows.append(segment_stats(prev, curr, sx, sy, True))
        keys = sorted({k for r in rows for k in r})
        print(f"{content}: {len(rows)} segments; share of the 1089 candidates a bound clears against the FINAL thresholds")
        for k in keys:
            v = [r[k] for r in rows if k in r]
            print(f"  {k:26s} mean {np.mean(v):10.4f}   min {np.min(v):10.4f}   max {np.max(v):10.4f}   (n={len(v)})")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
