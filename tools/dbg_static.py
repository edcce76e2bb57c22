"""Debug aid: the 'static with flat areas' adversarial frame through both motion modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linux_fg_amd import capi
rng = np.random.default_rng(77)
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 150)
for _ in range(4): rng.integers(0, 2, size=(h, w, 4), dtype=np.uint8)   # keep the generator in step with the test
rng = np.random.default_rng(77)
hi = rng.integers(254, 256, size=(h, w, 4), dtype=np.uint8); hi2 = rng.integers(254, 256, size=(h, w, 4), dtype=np.uint8)
lo = rng.integers(0, 2, size=(h, w, 4), dtype=np.uint8); lo2 = rng.integers(0, 2, size=(h, w, 4), dtype=np.uint8)
noise = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
stat = noise.copy(); stat[40:110, 30:150] = (90, 90, 90, 255); stat[:, w - 40:] = 0
ctx = capi.Context(0)
P, C = ctx.frame_from(stat), ctx.frame_from(stat.copy())
M = ctx.create_frame(w, h, capi.FORMAT_MV_S8X2)
out = {}
for mode in (capi.MOTION_EXACT_ONLY, capi.MOTION_PREFILTERED):
    ctx.set_motion_mode(mode); ctx.motion(P, C, M); ctx.sync(); out[mode] = ctx.download(M).copy()
bad = (out[0] != out[1]).any(-1)
print("differ:", int(bad.sum()), ctx.motion_last_stats())
ys, xs = np.nonzero(bad)
for y, x in list(zip(ys, xs))[:12]:
    print((x, y), "exact", out[1][y, x], "prefiltered", out[0][y, x], "tile", (x // 56, y // 64), "row in tile", y % 64, "col in tile", x % 56)
