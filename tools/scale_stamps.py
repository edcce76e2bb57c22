#!/usr/bin/env python3
"""Diagnostic: per-wave time stamps of scale_2x_kernel (needs a -DLFG_DIAG_STAMPS build: tools/build_scale_variant.sh stamps
-DLFG_DIAG_STAMPS; run with LFG_LIB=build_variants/lib_stamps.so).  Columns of the dump: wave, start, first rows in,
end of each step, (xcc << 32 | hw_id), (strip << 32 | column group); s_memrealtime ticks of 10 ns."""
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
tag = os.path.basename(os.environ.get("LFG_LIB", "default")).replace(".so", "")
out = os.path.join(ROOT, "gpurun_out", f"scale_stamps_{tag}.txt")
os.makedirs(os.path.dirname(out), exist_ok=True)
print("dump rc", ctx.lib.lfg_diag_dump_stamps(out.encode()))
d = np.loadtxt(out, dtype=np.uint64)
S = d.shape[1] - 5
nsteps = ((d[:, 4 + S] >> np.uint64(24)) & np.uint64(0xff)).astype(int)
t0 = d[:, 1].min()
ns = lambda col: (d[:, col] - t0).astype(np.float64) * 10.0
st, first, en = ns(1), ns(2), ns(2 + S)
q = lambda v: "min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max())
print("waves", len(d), "steps", S)
print("start        ns:", q(st))
print("first rows   ns:", q(first), "| wait p50 %.0f" % np.median(first - st))
print("end          ns:", q(en))
print("life         ns:", q(en - st))
for k in range(S):
    a0 = first if k == 0 else ns(2 + k)
    print(f"step {k} ns:", q(ns(3 + k) - a0))
for nn in sorted(set(nsteps.tolist())):
    m = nsteps == nn
    print(f"strips of {nn} steps: waves {m.sum()}, first rows p50 {np.median(first[m]):.0f}, end p50 {np.median(en[m]):.0f} p90 {np.percentile(en[m], 90):.0f} max {en[m].max():.0f}")
late = st > np.percentile(st, 75)
print("late starters (last quarter): start p50 %.0f, life p50 %.0f; early: life p50 %.0f" % (np.median(st[late]), np.median((en - st)[late]), np.median((en - st)[~late])))
clk = (d[:, 4 + S] >> np.uint64(32)).astype(np.float64)
ghz = clk / np.maximum(en - st, 10.0)
print("shader clock over wave life, GHz:", q(ghz * 1000) , "(x1e-3)")
xcc = (d[:, 3 + S] >> np.uint64(32)).astype(int)
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: waves {m.sum()}, end max {en[m].max():.0f}")
