"""Host-side logic that needs no GPU: the synthetic frame source (C++ vs Python), the exact 1/255
split used by the kernels, the golden fixtures against the oracle, and the sharding helpers."""
import os
import subprocess

import numpy as np
import pytest

from linux_fg_amd import sharding, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inv255_split_is_exact_for_every_byte():
    """csrc/lfg_device.hpp: fma(k, hi, k*lo) == (float)k / 255.0f for k = 0..255 (exact rationals)."""
    from fractions import Fraction
    hi = float.fromhex("0x1.010102p-8")
    lo = float.fromhex("-0x1.fdfdfep-33")
    assert np.float32(hi) == np.float32(1.0) / np.float32(255.0)
    for k in range(256):
        t = np.float32(k) * np.float32(lo)                       # one rounding
        exact = Fraction(k) * Fraction(hi) + Fraction(float(t))  # what the fma rounds
        got = np.float32(float(exact))                           # float(Fraction) is correctly rounded to double,
        want = np.float32(k) / np.float32(255.0)                 # and double -> float32 cannot double-round here
        lo_n, hi_n = np.nextafter(want, np.float32(-1)), np.nextafter(want, np.float32(2))
        assert abs(exact - Fraction(float(want))) <= min(abs(exact - Fraction(float(lo_n))), abs(exact - Fraction(float(hi_n))))
        assert got == want, k


def test_golden_fixtures_match_the_oracle(oracle):
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"))
    assert (oracle.scale(g["prev_in"], 128, 72) == g["prev_up"]).all()
    assert (oracle.scale(g["curr_in"], 128, 72) == g["curr_up"]).all()
    assert (oracle.scale(g["curr_in"], 53, 41) == g["curr_53x41"]).all()
    mv = oracle.motion(g["prev_up"], g["curr_up"], 8, 16.0)
    assert (mv.astype(np.int8) == g["mv"]).all()
    assert len(np.unique(g["mv"].reshape(-1, 2), axis=0)) > 3          # the fixture is not one flat vector field
    for t in (25, 50, 75):
        assert (oracle.interpolate(g["prev_up"], g["curr_up"], mv, t / 100.0) == g[f"interp_{t}"]).all()
    assert (oracle.motion(g["prev_in"], g["curr_in"], 4, 3.0).astype(np.int8) == g["mv_b4_r3"]).all()


@pytest.fixture(scope="module")
def synth_check(tmp_path_factory):
    """tests/cpp/synth_check.cpp: the C++ SyntheticCapture compiled on its own (no GPU code)."""
    out = tmp_path_factory.mktemp("cpp") / "synth_check"
    host = os.path.join(ROOT, "linux-fg_amd", "host")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", host, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "synth_check.cpp"),
                           os.path.join(host, "frame_source.cpp"), "-o", str(out)])
    return str(out)


@pytest.mark.parametrize("w,h,stream", [(64, 36, 0), (33, 17, 5)])
def test_cpp_synthetic_capture_equals_python(synth_check, w, h, stream):
    raw = subprocess.run([synth_check, str(w), str(h), str(stream), "3"], capture_output=True, check=True).stdout
    frames = np.frombuffer(raw, np.uint8).reshape(3, h, w, 4)
    seed = synth.BASE_SEED + stream
    want = synth.make_prev(w, h, seed)
    assert (frames[0] == want).all()
    for k in (1, 2):
        want = synth.translate(want, (3, -2), seed + k)
        assert (frames[k] == want).all()


def test_stream_assignment_covers_every_stream_once():
    for world in (1, 2, 4, 8):
        owned = [sharding.stream_of_rank(r, world) for r in range(world)]
        assert sorted(s for o in owned for s in o) == list(range(world))
        assert all(len(o) == 1 for o in owned)
    owned = [sharding.stream_of_rank(r, 3, streams=8) for r in range(3)]
    assert sorted(s for o in owned for s in o) == list(range(8))


def test_single_rank_broadcaster_is_a_pass_through():
    calls = []
    b = sharding.SharedFrameBroadcaster(1, None, world_size=1, refill=lambda step, s: calls.append((step, s)))
    b.start(0)
    for k in range(3):
        assert b.acquire(k) == 0                     # one slot, no transport, nothing to wait for
    b.drain()
    assert calls == [(0, 0), (1, 0), (2, 0)]
    with pytest.raises(ValueError):
        sharding.SharedFrameBroadcaster(2, None, world_size=2)      # several ranks need a transport


@pytest.mark.parametrize("in_h", [1, 2, 5, 16, 17, 36, 135, 540, 1080, 1081, 2160, 4320, 32768])
def test_scale_2x_strip_plan_covers_every_step_once(in_h):
    """csrc/scale.hip: scale_2x_strip_of (read through lfg_diag_scale_2x_strip; host arithmetic, no GPU): the
    in_h + 1 row steps 2 .. in_h + 2 of the 2x kernel fall into exactly one strip each, XCD bands are contiguous and in
    order, strips inside a band are contiguous, never longer than the kernel's unrolled length and never grow with
    the dispatch index (longest first)."""
    import ctypes
    from linux_fg_amd import capi
    lib = capi.load()
    per, first, steps = ctypes.c_uint32(), ctypes.c_int32(), ctypes.c_int32()
    assert lib.lfg_diag_scale_2x_strip(in_h, 0, 0, ctypes.byref(per), ctypes.byref(first), ctypes.byref(steps)) == 0
    assert per.value % 3 == 0 and per.value >= 3
    nxt, longest = 2, 8                                             # (the kernel is unrolled to at most 8 steps)
    for x in range(8):
        last_len = None
        for i in range(per.value):
            assert lib.lfg_diag_scale_2x_strip(in_h, x, i, None, ctypes.byref(first), ctypes.byref(steps)) == 0
            if steps.value == 0:
                continue
            assert first.value == nxt and 0 < steps.value <= longest
            assert last_len is None or steps.value <= last_len
            last_len = steps.value
            nxt += steps.value
    assert nxt == in_h + 3                                          # one past the last step, in_h + 2
    assert lib.lfg_diag_scale_2x_strip(in_h, 8, 0, None, None, None) != 0
    assert lib.lfg_diag_scale_2x_strip(in_h, 0, per.value, None, None, None) != 0
