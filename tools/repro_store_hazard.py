#!/usr/bin/env python3
"""On the GPU box: does the 16-byte store of the 2x scale kernel need its wait states (csrc/lfg_device.hpp:
store_b128_guarded)?  Runs the 1080p -> 4K upscale LFG_REPRO_RUNS times with the library named by LFG_LIB and compares
every output frame, byte for byte, with the first one; prints the runs and pixels that differ.  Build the unguarded
library with  LFG_SKIP_HAZARD_CHECK=1 tools/build_scale_variant.sh noguard -DLFG_DIAG_NO_STORE_GUARD  and run both."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from linux_fg_amd import capi, synth  # noqa: E402

runs = int(os.environ.get("LFG_REPRO_RUNS", "600"))
ctx = capi.Context(0)
w, h = 1920, 1080
src = ctx.frame_from(synth.make_prev(w, h, synth.BASE_SEED))
outs = [ctx.create_frame(2 * w, 2 * h) for _ in range(4)]
ctx.scale(src, outs[0]); ctx.sync()
ref = ctx.download(outs[0]).copy()
bad_runs = bad_pixels = 0
for r in range(0, runs, 4):
    for o in outs:                       # four launches back to back, as the benchmark issues them
        ctx.scale(src, o)
    ctx.sync()
    for o in outs:
        d = int((ctx.download(o) != ref).any(-1).sum())
        bad_runs += d != 0
        bad_pixels += d
print(f"{os.environ.get('LFG_LIB', 'default library')}: {runs} launches, {bad_runs} with a pixel that differs from the first launch's, {bad_pixels} such pixels")
