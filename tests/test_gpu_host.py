"""The C++ host mirror (linux-fg_amd/host: Scaler / FrameManager / HipContext, built as lfg_host)
driven end to end on the GPU and checked against the oracle: ProcessFrame's presentation order
(real, then [generated, real] per further input frame), the upscaled frames (+-1 LSB) and the
generated frames (exact, from the device's own upscaled frames)."""
import json
import os
import subprocess

import numpy as np
import pytest

from linux_fg_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "linux-fg_amd", "lfg_host")


@pytest.fixture(scope="module")
def host_binary():
    if not os.path.exists(HOST):
        import __graft_entry__ as entry
        entry.build()
    return HOST


def test_process_frame_sequence_matches_oracle(host_binary, oracle, tmp_path):
    w, h, n = 64, 36, 3
    out = subprocess.run([host_binary, "--input-width", str(w), "--input-height", str(h), "--output-width", str(2 * w),
                          "--output-height", str(2 * h), "--frames", str(n), "--dump-dir", str(tmp_path), "--quiet", "2"],
                         capture_output=True, text=True, check=True)
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["presented"] == 2 * n - 1 and stats["interpolated"] == n - 1
    files = sorted(os.listdir(tmp_path))
    kinds = [f.split("_")[2] for f in files]
    assert kinds == ["real", "interp", "real", "interp", "real"]         # presentation order
    frames = [np.fromfile(os.path.join(tmp_path, f), np.uint8).reshape(2 * h, 2 * w, 4) for f in files]
    # inputs as the C++ SyntheticCapture generates them (stream 2)
    seed = synth.BASE_SEED + 2
    inputs = [synth.make_prev(w, h, seed)]
    for k in range(1, n):
        inputs.append(synth.translate(inputs[-1], (3, -2), seed + k))
    reals = [frames[0], frames[2], frames[4]]
    for got, src in zip(reals, inputs):
        d = np.abs(got.astype(np.int16) - oracle.scale(src, 2 * w, 2 * h).astype(np.int16))
        assert d.max() <= 1
    for i, gen in enumerate([frames[1], frames[3]]):
        prev_up, curr_up = reals[i], reals[i + 1]
        mv = oracle.motion(prev_up, curr_up, 8, 16.0)
        assert (gen == oracle.interpolate(prev_up, curr_up, mv, 0.5)).all()


def test_no_interpolation_and_aspect_ratio(host_binary):
    out = subprocess.run([host_binary, "--input-width", "80", "--input-height", "40", "--output-height", "100",
                          "--no-interpolation", "--frames", "4", "--quiet"], capture_output=True, text=True, check=True)
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["presented"] == 4 and stats["interpolated"] == 0      # output width derived: 200 (src/main.cpp:76-90)
