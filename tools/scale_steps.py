#!/usr/bin/env python3
"""Diagnostic: per-step s_memtime stamps of scale_2x_kernel (-DLFG_DIAG_STAMPS build)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from linux_fg_amd import capi, synth
ctx = capi.Context(0)
a = synth.make_prev(1920, 1080)
fi = ctx.frame_from(a); fo = ctx.create_frame(3840, 2160)
for _ in range(5):
    ctx.scale(fi, fo)
ctx.sync()
ctx.lib.lfg_diag_dump_stamps.argtypes = [ctypes.c_char_p]
out = os.path.join(ROOT, "gpurun_out", "scale_steps.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
ctx.lib.lfg_diag_dump_stamps(out.encode())
d = np.loadtxt(out, dtype=np.uint64)
steps = d[:, 4:].astype(np.int64)
dt = np.diff(steps, axis=1)
print("waves", len(d), "steps", steps.shape[1])
print("median cycles per step:", np.median(dt, axis=0).astype(int).tolist())
print("p90    cycles per step:", np.percentile(dt, 90, axis=0).astype(int).tolist())
print("total loop cycles p50 %d" % np.median(steps[:, -1] - steps[:, 0]))
life = (d[:, 3] - d[:, 1]).astype(np.float64) * 10
print("lifetime ns p50 %.0f max %.0f; prologue ns p50 %.0f" % (np.median(life), life.max(), np.median((d[:, 2] - d[:, 1]).astype(np.float64) * 10)))
