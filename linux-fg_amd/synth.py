"""Synthetic frame source: the headless stand-in for the reference's X11 window capture
(/root/reference/src/window_capture.cpp:232-470 is out of scope; the GPU box has no display).

Content model: ``prev`` is a low-frequency gradient plus uniform
noise in every channel (alpha included) so that block-match costs have unique minima;
``curr`` is ``prev`` translated by a known integer vector with the exposed border filled
from a second noise stream.  The generator is a counter-based 32-bit hash (the
"lowbias32" integer finaliser) of (seed, pixel index, channel), so the same frame can be
produced from numpy here and from C++ in host/synthetic_capture.hpp.
Stream ``i`` uses seed 0x5EED0000 + i.
"""
from __future__ import annotations

import numpy as np

BASE_SEED = 0x5EED0000
DEFAULT_SHIFT = (3, -2)


def _lowbias32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7FEB352D)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846CA68B)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def noise_bytes(width: int, height: int, seed: int) -> np.ndarray:
    """(H, W, 4) uint8, one hash per pixel, one byte of it per channel."""
    idx = np.arange(width * height, dtype=np.uint32).reshape(height, width)
    with np.errstate(over="ignore"):
        h = _lowbias32(idx + _lowbias32(np.array([seed & 0xFFFFFFFF], np.uint32))[0])
    out = np.empty((height, width, 4), np.uint8)
    for c in range(4):
        out[..., c] = (h >> np.uint32(8 * c)) & np.uint32(0xFF)
    return out


def make_prev(width: int, height: int, seed: int = BASE_SEED) -> np.ndarray:
    """Gradient + noise.  byte = ((x*(c+1) + 2*y) >> 3) + (noise & 0x7F), wrapped to 8 bits."""
    x = np.arange(width, dtype=np.uint32)[None, :, None]
    y = np.arange(height, dtype=np.uint32)[:, None, None]
    c = np.arange(4, dtype=np.uint32)[None, None, :]
    grad = (x * (c + 1) + 2 * y) >> 3
    n = noise_bytes(width, height, seed).astype(np.uint32) & 0x7F
    return ((grad + n) & 0xFF).astype(np.uint8)


def translate(prev: np.ndarray, shift=DEFAULT_SHIFT, seed: int = BASE_SEED) -> np.ndarray:
    """curr(q) = prev(q - shift); pixels with no source come from a second noise stream."""
    H, W = prev.shape[:2]
    tx, ty = int(shift[0]), int(shift[1])
    curr = noise_bytes(W, H, (seed ^ 0xA5A5A5A5) & 0xFFFFFFFF)
    ys0, ys1 = max(0, ty), min(H, H + ty)
    xs0, xs1 = max(0, tx), min(W, W + tx)
    if ys1 > ys0 and xs1 > xs0:
        curr[ys0:ys1, xs0:xs1] = prev[ys0 - ty:ys1 - ty, xs0 - tx:xs1 - tx]
    return curr


def make_pair(width: int, height: int, stream: int = 0, shift=DEFAULT_SHIFT):
    """(prev, curr) for stream ``stream``: curr is prev translated by ``shift``."""
    seed = (BASE_SEED + stream) & 0xFFFFFFFF
    prev = make_prev(width, height, seed)
    return prev, translate(prev, shift, seed)


def make_uncorrelated_pair(width: int, height: int, stream: int = 0):
    """Two independent noise frames: worst case for block matching (no good match anywhere)."""
    seed = (BASE_SEED + stream) & 0xFFFFFFFF
    return noise_bytes(width, height, seed), noise_bytes(width, height, (seed * 2654435761 + 1) & 0xFFFFFFFF)
