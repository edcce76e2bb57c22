#!/usr/bin/env python3
"""How many segments the lean kernel (csrc/motion_lean.hip) settles on each synthetic content (LFG_DEBUG prints the counters)."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LFG_DEBUG"] = "1"
from linux_fg_amd import capi
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
ctx = capi.Context(0)
ctx.lanes(2)
for content in (sys.argv[1:] or bench.CONTENTS):
    p_in, c_in = bench.make_content(content, 1920, 1080, 0, 0)
    p, c = ctx.frame_from(p_in), ctx.frame_from(c_in)
    P, C = ctx.create_frame(3840, 2160), ctx.create_frame(3840, 2160)
    M = ctx.create_frame(3840, 2160, capi.FORMAT_MV_S8X2)
    ctx.scale(p, P); ctx.scale(c, C); ctx.motion(P, C, M); ctx.sync()
    print("==", content, flush=True)
    ctx.motion_last_stats()
    for f in (p, c, P, C, M):
        ctx.destroy_frame(f)
