#!/usr/bin/env python3
"""How many 16-row segments the prefilter leaves to the resolve kernel on each synthetic content at 1080p -> 4K (lfg_motion_open_segments),
with one frame at a time and with frames in flight."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linux_fg_amd import capi
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
for lanes in (1, 3):
    ctx = capi.Context(0)
    if lanes > 1:
        ctx.lanes(lanes)
    for content in (sys.argv[1:] or bench.CONTENTS):
        p_in, c_in = bench.make_content(content, 1920, 1080, 0, 0)
        p, c = ctx.frame_from(p_in), ctx.frame_from(c_in)
        P, C = ctx.create_frame(3840, 2160), ctx.create_frame(3840, 2160)
        M = ctx.create_frame(3840, 2160, capi.FORMAT_MV_S8X2)
        ctx.scale(p, P); ctx.scale(c, C)
        for _ in range(3):
            ctx.motion(P, C, M)
        ctx.sync()
        print(f"lanes {lanes} {content:13s} open segments {ctx.motion_open_segments()}  stats {ctx.motion_last_stats()}", flush=True)
        for f in (p, c, P, C, M):
            ctx.destroy_frame(f)
    ctx.close()
