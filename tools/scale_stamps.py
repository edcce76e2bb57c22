#!/usr/bin/env python3
"""Diagnostic: per-wave start/end stamps of scale_2x_kernel (needs a -DLFG_DIAG_STAMPS build of the library)."""
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
out = os.path.join(ROOT, "gpurun_out", "scale_stamps.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
print("dump rc", ctx.lib.lfg_diag_dump_stamps(out.encode()))
d = np.loadtxt(out, dtype=np.uint64)
t0 = d[:, 1].min()
st = (d[:, 1] - t0).astype(np.float64) * 10.0    # s_memrealtime ticks at 100 MHz -> ns
pro = (d[:, 2] - t0).astype(np.float64) * 10.0
en = (d[:, 3] - t0).astype(np.float64) * 10.0
print("waves", len(d))
print("start  ns: min %.0f p50 %.0f p90 %.0f max %.0f" % (st.min(), np.median(st), np.percentile(st, 90), st.max()))
print("prolog ns: p50 %.0f max %.0f (duration p50 %.0f)" % (np.median(pro), pro.max(), np.median(pro - st)))
print("end    ns: min %.0f p50 %.0f p90 %.0f max %.0f" % (en.min(), np.median(en), np.percentile(en, 90), en.max()))
print("life   ns: min %.0f p50 %.0f max %.0f" % ((en - st).min(), np.median(en - st), (en - st).max()))
