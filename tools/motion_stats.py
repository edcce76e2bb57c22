import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linux_fg_amd import capi, synth
ctx = capi.Context(0)
w, h = 1920, 1080
W, H = 2 * w, 2 * h
prev_in = synth.make_prev(w, h, synth.BASE_SEED)
curr_in = synth.translate(prev_in, (3, -2), synth.BASE_SEED)
p, c = ctx.frame_from(prev_in), ctx.frame_from(curr_in)
P, C = ctx.create_frame(W, H), ctx.create_frame(W, H)
M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
ctx.scale(p, P); ctx.scale(c, C)
ctx.motion(P, C, M)
print("pipeline content (upscaled frames):", ctx.motion_last_stats())
mv = ctx.download(M)
vals, counts = np.unique(mv.reshape(-1, 2), axis=0, return_counts=True)
order = np.argsort(-counts)[:6]
print("most common MVs:", [(tuple(vals[i]), int(counts[i])) for i in order])
