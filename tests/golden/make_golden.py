#!/usr/bin/env python3
"""Generates tests/golden/golden_small.npz from the CPU oracle (oracle/lfg_oracle.c).

The reference has no golden vectors of its own and cannot run in this image, so
these fixtures pin the ORACLE'S OWN behaviour, not the reference's: they catch accidental changes
of the oracle and give the GPU tests a fixed target that does not depend on rebuilding it.
Inputs: 64x36 seeded synthetic frames (linux-fg_amd/synth.py).  Outputs: scale to 128x72,
motion (blockSize 8, searchRadius 16) and interpolate (t = 0.25, 0.5, 0.75) on the 128x72 frames.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from linux_fg_amd import synth  # noqa: E402

w, h = 64, 36
prev_in, curr_in = synth.make_pair(w, h, stream=0, shift=(2, -1))
curr_in = curr_in.copy()
curr_in[10:20, 30:44] = synth.noise_bytes(14, 10, 4242)          # an occluder: non-trivial vectors
prev_up = oracle.scale(prev_in, 2 * w, 2 * h)
curr_up = oracle.scale(curr_in, 2 * w, 2 * h)
mv = oracle.motion(prev_up, curr_up, 8, 16.0)
out = {"prev_in": prev_in, "curr_in": curr_in, "prev_up": prev_up, "curr_up": curr_up, "mv": mv.astype(np.int8)}
for t in (0.25, 0.5, 0.75):
    out[f"interp_{int(t * 100)}"] = oracle.interpolate(prev_up, curr_up, mv, t)
# a generic-ratio scale and a non-default motion parameter set
out["curr_53x41"] = oracle.scale(curr_in, 53, 41)
out["mv_b4_r3"] = oracle.motion(prev_in, curr_in, 4, 3.0).astype(np.int8)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")
np.savez_compressed(path, **out)
print("wrote", path, {k: v.shape for k, v in out.items()})
