"""oracle/lfg_oracle.c against the independently written numpy restatement: bit-for-bit.

Two separately written restatements of the same shader text agreeing exactly is the
strongest pin available while the reference itself cannot run here.  CPU only.
"""
import numpy as np
import pytest

from linux_fg_amd import synth
from oracle import numpy_restatement as npr

RNG = np.random.default_rng(99)


def rand_frame(w, h):
    return RNG.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


@pytest.mark.parametrize("in_wh,out_wh", [((16, 9), (32, 18)), ((11, 7), (29, 16)), ((20, 12), (13, 8)),
                                           ((8, 8), (8, 8)), ((5, 3), (17, 4))])
def test_scale_c_equals_numpy(oracle, in_wh, out_wh):
    f = rand_frame(*in_wh)
    a = oracle.scale(f, *out_wh)
    b = npr.scale(f, *out_wh)
    assert (a == b).all()


@pytest.mark.parametrize("bs,R", [(8, 16.0), (4, 3.0), (5, 2.0), (8, 0.0)])
def test_motion_c_equals_numpy(oracle, bs, R):
    W, H = (26, 22) if R > 8 else (21, 15)
    prev = synth.make_prev(W, H, seed=3)
    curr = synth.translate(prev, (2, -1), seed=3)
    curr[::3, ::4] = RNG.integers(0, 256, size=curr[::3, ::4].shape, dtype=np.uint8)  # break exact matches
    a = oracle.motion(prev, curr, bs, R)
    b = npr.motion(prev, curr, bs, R)
    assert (a == b).all()


def test_motion_uncorrelated_c_equals_numpy(oracle):
    prev, curr = synth.make_uncorrelated_pair(20, 12, stream=1)
    assert (oracle.motion(prev, curr, 8, 4.0) == npr.motion(prev, curr, 8, 4.0)).all()


@pytest.mark.parametrize("t", [0.25, 0.5, 0.75, 0.3])
def test_interpolate_c_equals_numpy(oracle, t):
    W, H = 48, 20
    p, c = rand_frame(W, H), rand_frame(W, H)
    mv = RNG.integers(-2, 3, size=(H, W, 2)).astype(np.float32)
    mv[RNG.random((H, W)) < 0.5] = 0.0
    a = oracle.interpolate(p, c, mv, t)
    b = npr.interpolate(p, c, mv, t)
    assert (a == b).all()


def test_interpolate_non_pow2_width_c_equals_numpy(oracle):
    W, H = 30, 18       # uv*W - 0.5 is not exact here: exercises the fp32 drift of the bilinear weights
    p, c = rand_frame(W, H), rand_frame(W, H)
    mv = np.zeros((H, W, 2), np.float32)
    assert (oracle.interpolate(p, c, mv, 0.5) == npr.interpolate(p, c, mv, 0.5)).all()
