import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes, numpy as np
from linux_fg_amd import capi, synth
ctx = capi.Context(0)
fi = ctx.frame_from(synth.make_prev(1920, 1080)); fo = ctx.create_frame(3840, 2160)
for _ in range(5): ctx.scale(fi, fo)
ctx.sync()
ctx.lib.lfg_diag_dump_stamps.argtypes = [ctypes.c_char_p]
out = os.path.join(ROOT, "gpurun_out", "scale_sub.txt"); os.makedirs(os.path.dirname(out), exist_ok=True)
ctx.lib.lfg_diag_dump_stamps(out.encode())
d = np.loadtxt(out, dtype=np.uint64)[:, 4:12].astype(np.int64)
names = ["s5: start->LDS written", "s5: LDS+horizontal", "s5: vertical+pack", "s5: exchange+store", "s10: start->LDS written", "s10: LDS+horizontal", "s10: vertical+pack", "s10: exchange+store"]
for i, n in enumerate(names): print("%-26s median %6d  p90 %6d" % (n, np.median(d[:, i]), np.percentile(d[:, i], 90)))
