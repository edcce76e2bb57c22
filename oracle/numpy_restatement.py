"""Independent numpy restatement of the three shaders, for cross-checking lfg_oracle.c.

TEST INFRASTRUCTURE ONLY (same rules as lfg_oracle.c).  Written separately from the C
file, array-at-a-time instead of pixel-at-a-time, but with the same fp32 operation
order, so the two must agree bit for bit.  numpy float32 arithmetic is IEEE single
with one rounding per operation and no FMA contraction.

Follows /root/reference/shaders/scale.comp:14-61, motion.comp:16-57,
interpolate.comp:15-40 and the nine written choices in lfg_oracle.c's header.
Only meant for small frames (it loops over taps / candidates in Python).
"""
from __future__ import annotations

import numpy as np

F = np.float32


def _to_float(img: np.ndarray) -> np.ndarray:
    return img.astype(F) / F(255.0)


def _to_unorm8(v: np.ndarray) -> np.ndarray:
    v = np.where(v > F(0.0), v, F(0.0)).astype(F)
    v = np.where(v > F(1.0), F(1.0), v).astype(F)
    return np.rint(v * F(255.0)).astype(np.uint8)      # np.rint rounds half to even


def _bilinear(imgf: np.ndarray, s: np.ndarray, t: np.ndarray) -> np.ndarray:
    H, W = imgf.shape[:2]
    u = s * F(W) - F(0.5)
    v = t * F(H) - F(0.5)
    fu, fv = np.floor(u), np.floor(v)
    a, b = (u - fu).astype(F), (v - fv).astype(F)
    i0, j0 = fu.astype(np.int64), fv.astype(np.int64)
    i1, j1 = i0 + 1, j0 + 1
    i0, i1 = np.clip(i0, 0, W - 1), np.clip(i1, 0, W - 1)
    j0, j1 = np.clip(j0, 0, H - 1), np.clip(j1, 0, H - 1)
    one = F(1.0)
    w00 = ((one - a) * (one - b))[..., None]
    w10 = (a * (one - b))[..., None]
    w01 = ((one - a) * b)[..., None]
    w11 = (a * b)[..., None]
    return ((w00 * imgf[j0, i0] + w10 * imgf[j0, i1]) + w01 * imgf[j1, i0]) + w11 * imgf[j1, i1]


def _sin(x: np.ndarray) -> np.ndarray:
    return np.sin(x.astype(np.float64)).astype(F)


def _lanczos(x: np.ndarray) -> np.ndarray:
    A = F(3.0)
    px = F(3.14159265359) * x
    with np.errstate(divide="ignore", invalid="ignore"):
        r = A * _sin(px) * _sin(px / A) / (px * px)
    return np.where(x == F(0.0), F(1.0), r).astype(F)


def scale(frame: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    inH, inW = frame.shape[:2]
    imgf = _to_float(frame)
    px = np.arange(out_w, dtype=F)[None, :].repeat(out_h, 0)
    py = np.arange(out_h, dtype=F)[:, None].repeat(out_w, 1)
    uvx = (px + F(0.5)) / F(out_w)
    uvy = (py + F(0.5)) / F(out_h)
    tsx, tsy = F(1.0) / F(inW), F(1.0) / F(inH)
    ppx = uvx * F(inW) - F(0.5)
    ppy = uvy * F(inH) - F(0.5)
    fx, fy = ppx - np.floor(ppx), ppy - np.floor(ppy)
    sx, sy = np.floor(ppx) - F(2.0), np.floor(ppy) - F(2.0)
    color = np.zeros((out_h, out_w, 4), F)
    total = np.zeros((out_h, out_w), F)
    for y in range(6):
        for x in range(6):
            spx = (sx + F(x) + F(0.5)) * tsx
            spy = (sy + F(y) + F(0.5)) * tsy
            skip = (spx < 0) | (spy < 0) | (spx > 1) | (spy > 1)
            weight = _lanczos(F(x) - fx - F(2.0)) * _lanczos(F(y) - fy - F(2.0))
            tex = _bilinear(imgf, np.clip(spx, 0, 1).astype(F), np.clip(spy, 0, 1).astype(F))
            color = np.where(skip[..., None], color, color + tex * weight[..., None]).astype(F)
            total = np.where(skip, total, total + weight).astype(F)
    return _to_unorm8(color / total[..., None])


def motion(prev: np.ndarray, curr: np.ndarray, block_size: int = 8, search_radius: float = 16.0) -> np.ndarray:
    H, W = curr.shape[:2]
    R = int(search_radius)
    assert float(R) == float(search_radius), "integer radii only in this restatement"
    bs = int(block_size)
    pad = bs + R + 1
    currf = np.zeros((H + 2 * pad, W + 2 * pad, 4), F)
    prevf = np.zeros_like(currf)                         # out-of-bounds prev fetch -> 0
    currf[pad:pad + H, pad:pad + W] = _to_float(curr)
    prevf[pad:pad + H, pad:pad + W] = _to_float(prev)
    inb = np.zeros((H + 2 * pad, W + 2 * pad), bool)
    inb[pad:pad + H, pad:pad + W] = True
    best = np.zeros((H, W, 2), F)
    mind = np.full((H, W), F(1e10), F)
    half = bs // 2
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            diff = np.zeros((H, W), F)
            for y in range(bs):
                for x in range(bs):
                    oy, ox = pad - half + y, pad - half + x       # block position of pixel (0,0)
                    c = currf[oy:oy + H, ox:ox + W]
                    p = prevf[oy + dy:oy + dy + H, ox + dx:ox + dx + W]
                    d = c - p
                    sq = d * d
                    dist = np.sqrt(((sq[..., 0] + sq[..., 1]) + sq[..., 2]) + sq[..., 3]).astype(F)
                    diff = np.where(inb[oy:oy + H, ox:ox + W], diff + dist, diff).astype(F)
            better = diff < mind
            mind = np.where(better, diff, mind)
            best[better] = (F(dx), F(dy))
    return best


def interpolate(prev: np.ndarray, curr: np.ndarray, mv: np.ndarray, factor: float = 0.5) -> np.ndarray:
    H, W = curr.shape[:2]
    t = F(factor)
    px = np.arange(W, dtype=F)[None, :].repeat(H, 0)
    py = np.arange(H, dtype=F)[:, None].repeat(W, 1)
    uvx = (px + F(0.5)) / F(W)
    uvy = (py + F(0.5)) / F(H)
    mx, my = mv[..., 0].astype(F), mv[..., 1].astype(F)

    def sample(img, scale):
        sx = uvx + mx * scale
        sy = uvy + my * scale
        out = (sx < 0) | (sy < 0) | (sx > 1) | (sy > 1)
        tex = _bilinear(_to_float(img), np.clip(sx, 0, 1).astype(F), np.clip(sy, 0, 1).astype(F))
        return np.where(out[..., None], F(0.0), tex).astype(F)

    p = sample(prev, -t)
    c = sample(curr, F(1.0) - t)
    return _to_unorm8(p * (F(1.0) - t) + c * t)
