"""Debug aid: a frame pair on which every tile overflows its lists (flat frames one level apart with a sparse dot
pattern that keeps the windows from being one colour): time of the prefiltered path against the literal kernel alone."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linux_fg_amd import capi
ctx = capi.Context(0)
W, H = 3840, 2160
prev = np.full((H, W, 4), 100, np.uint8); curr = np.full((H, W, 4), 101, np.uint8)
prev[::29, ::31] = 103; curr[::29, ::31] = 104
P, C = ctx.frame_from(prev), ctx.frame_from(curr); M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
out = {}
for mode in (capi.MOTION_PREFILTERED, capi.MOTION_EXACT_ONLY):
    ctx.set_motion_mode(mode)
    ctx.motion(P, C, M); ctx.sync()
    t = time.time()
    for _ in range(3): ctx.motion(P, C, M)
    ctx.sync()
    out[mode] = ctx.download(M).copy()
    print("mode", mode, round((time.time() - t) / 3 * 1e3, 2), "ms", ctx.motion_last_stats()[:2] if mode == 0 else "")
print("differing", int((out[0] != out[1]).any(-1).sum()))
