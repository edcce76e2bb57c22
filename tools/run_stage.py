#!/usr/bin/env python3
"""Run one stage of the hot path a few times on cuda:0 (for rocprofv3 --pmc / --kernel-trace runs).
usage: run_stage.py {scale|motion|interpolate|pipeline} [reps] [content]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from linux_fg_amd import capi, synth  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "pipeline"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
content = sys.argv[3] if len(sys.argv) > 3 else "translated"
w, h = (int(v) for v in os.environ.get("LFG_STAGE_INPUT", "1920x1080").split("x"))
W, H = 2 * w, 2 * h
ctx = capi.Context(0)
lanes = int(os.environ.get("LFG_STAGE_LANES", "1"))       # frames in flight (lanes of the C-ABI): the pipeline stage only
if lanes > 1:
    ctx.lanes(lanes)
import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
_bench = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_bench)
prev_in, curr_in = _bench.make_content(content, w, h, 0, 0)          # the benchmark's own frames
p_in, c_in = ctx.frame_from(prev_in), ctx.frame_from(curr_in)
P, C, O = ctx.create_frame(W, H), ctx.create_frame(W, H), ctx.create_frame(W, H)
M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
ctx.scale(p_in, P)
ctx.scale(c_in, C)
ctx.motion(P, C, M) if stage in ("interpolate",) else None
ctx.sync()
if lanes > 1 and stage == "pipeline":
    # step k on lane k % lanes with its own buffers, as bench.py runs them (the previous step's upscale is waited for)
    bufs = [(C, M, O)] + [(ctx.create_frame(W, H), ctx.create_frame(W, H, capi.FORMAT_MV_S8X2), ctx.create_frame(W, H)) for _ in range(lanes - 1)]
    for k in range(reps):
        ctx.lane_select(k % lanes)
        ctx.lane_wait((k - 1) % lanes)
        c4, m4, o4 = bufs[k % lanes]
        ctx.scale(c_in, c4)
        ctx.lane_mark()
        ctx.motion(P, c4, m4)
        ctx.interpolate(P, c4, m4, o4, 0.5)
    ctx.lane_select(0)
else:
    for _ in range(reps):
        if stage in ("scale", "pipeline"):
            ctx.scale(c_in, C)
        if stage in ("motion", "pipeline"):
            ctx.motion(P, C, M)
        if stage in ("interpolate", "pipeline"):
            ctx.interpolate(P, C, M, O, 0.5)
ctx.sync()
print("done", stage, reps)
