"""Debug aid: the benchmark's frames (translated 1080p pair, both upscaled on the device) through lfg_motion; prints
the fallback statistics (set LFG_DEBUG=1 for the list of flagged tiles) and compares with the literal kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from linux_fg_amd import capi, synth
ctx = capi.Context(0)
W, H = 3840, 2160
pin = synth.make_prev(W // 2, H // 2, seed=synth.BASE_SEED); cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
Pin, Cin = ctx.frame_from(pin), ctx.frame_from(cin)
P, C = ctx.create_frame(W, H), ctx.create_frame(W, H); M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
ctx.scale(Pin, P); ctx.scale(Cin, C)
out = {}
for mode in (capi.MOTION_EXACT_ONLY, capi.MOTION_PREFILTERED):
    ctx.set_motion_mode(mode); ctx.motion(P, C, M); ctx.sync(); out[mode] = ctx.download(M).copy()
print("stats", ctx.motion_last_stats(), "differing", int((out[0] != out[1]).any(-1).sum()))
