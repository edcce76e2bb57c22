"""Debug aid: does the 1080p->4K scale output hold anything but the oracle's pixels?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from linux_fg_amd import capi, synth
ctx = capi.Context(0)
W, H = 3840, 2160
pin = synth.make_prev(W // 2, H // 2, seed=synth.BASE_SEED); cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
for name, src in (("prev", pin), ("curr", cin)):
    want = oracle.scale(src, W, H).astype(np.int16)
    S = ctx.frame_from(src); D = ctx.create_frame(W, H)
    for rep in range(3):
        ctx.scale(S, D); ctx.sync()
        got = ctx.download(D).astype(np.int16)
        bad = np.argwhere(np.abs(got - want).max(-1) > 1)
        print(name, "rep", rep, "pixels off by more than 1 LSB:", len(bad), bad[:6].tolist())
        if len(bad):
            y, x = bad[0]; print("  got", got[y, x], "want", want[y, x])
