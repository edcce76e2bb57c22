"""Debug aid: prefiltered vs exact-only motion on the same frames; prints where they differ."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linux_fg_amd import capi, synth

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ctx = capi.Context(0)
prev = synth.make_prev(W, H, seed=synth.BASE_SEED)
curr = synth.translate(prev, (3, -2), synth.BASE_SEED)
P, C = ctx.frame_from(prev), ctx.frame_from(curr)
M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
out = {}
for mode in (capi.MOTION_EXACT_ONLY, capi.MOTION_PREFILTERED):
    ctx.set_motion_mode(mode)
    ctx.motion(P, C, M)
    ctx.sync()
    out[mode] = ctx.download(M).copy()
print("stats", ctx.motion_last_stats())
bad = (out[0] != out[1]).any(-1)
print("differing pixels:", int(bad.sum()), "of", bad.size)
if bad.any():
    ys, xs = np.nonzero(bad)
    print("x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
    print("x mod 56 histogram", np.bincount(xs % 56, minlength=56))
    print("y mod 64 histogram", np.bincount(ys % 64, minlength=64))
    for k in range(min(8, len(ys))):
        y, x = ys[k], xs[k]
        print((x, y), "exact", out[1][y, x], "prefiltered", out[0][y, x])
