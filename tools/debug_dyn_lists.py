#!/usr/bin/env python3
"""The hand-over test's frame (tests/test_gpu_parity.py: test_motion_hand_over_queue_overflows_gracefully) through lfg_motion with
LFG_DEBUG_DYN=1: depth histogram of the handed-over parts' private lists and the records of the deepest ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from linux_fg_amd import capi, synth
W, H = 3840, 2160
prev = synth.make_prev(W, H, seed=synth.BASE_SEED + 9)
curr = synth.noise_bytes(W, H, 777)
moved = synth.translate(prev, (5, 3), synth.BASE_SEED + 9)
for gy in range(16):
    for gx in range(16):
        cx, cy = (2 * gx + 1) * W // 32, (2 * gy + 1) * H // 32
        curr[cy - 14:cy + 14, cx - 14:cx + 14] = moved[cy - 14:cy + 14, cx - 14:cx + 14]
ctx = capi.Context(0)
P, C = ctx.frame_from(prev), ctx.frame_from(curr)
M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
ctx.motion(P, C, M)
ctx.sync()
print(ctx.motion_last_stats())
