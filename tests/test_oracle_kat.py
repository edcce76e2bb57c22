"""Known-answer tests that pin the CPU oracle (oracle/lfg_oracle.c).

The reference has no tests or golden vectors, so these analytic cases,
derived from the shader text in SURVEY.md section 8(c), are what the oracle is pinned by.
CPU only.
"""
import numpy as np
import pytest

from linux_fg_amd import synth

RNG = np.random.default_rng(1234)


def rand_frame(w, h):
    return RNG.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


# ------------------------------------------------------------------ scale.comp

def test_s_kat1_constant_colour(oracle):
    """Constant input -> identical constant output everywhere, borders included (renormalisation)."""
    f = np.empty((9, 13, 4), np.uint8)
    f[...] = (10, 200, 77, 255)
    for (ow, oh) in [(26, 18), (13, 9), (31, 20), (7, 5)]:
        out = oracle.scale(f, ow, oh)
        assert (out == np.array([10, 200, 77, 255], np.uint8)).all(), (ow, oh)


def test_s_kat2_identity_size(oracle):
    """outSize == inSize -> f = 0 -> weights (0,0,1,0,0,0) up to ~1e-8 -> output == input."""
    f = rand_frame(37, 23)
    assert (oracle.scale(f, 37, 23) == f).all()


def test_s_kat3_phase_weights_at_2x(oracle):
    """Exact 2x: two phases per axis; raw weights as listed in SURVEY.md section 8(a) S1."""
    expect = np.array([0.030021, -0.132871, 0.890067, 0.270190, -0.067791, 0.007356], np.float32)
    for n_in in (960, 1920, 3840):
        s_even, w_even = oracle.lanczos_taps(2 * 100, n_in, 2 * n_in)      # f = 0.75
        s_odd, w_odd = oracle.lanczos_taps(2 * 100 + 1, n_in, 2 * n_in)    # f = 0.25
        assert s_even == 100 - 3 and s_odd == 100 - 2
        assert np.allclose(w_odd, expect, atol=3e-4)
        assert np.allclose(w_even, expect[::-1], atol=3e-4)
        assert abs(float(w_odd.sum()) - 0.996971) < 5e-4


def test_s_kat3_impulse_is_outer_product(oracle):
    """A single white texel on black at 2x gives the outer product of the two phase vectors,
    renormalised by the tap-weight sum and clamped at 0 for the negative lobes."""
    n = 16
    f = np.zeros((n, n, 4), np.uint8)
    f[8, 8] = 255
    out = oracle.scale(f, 2 * n, 2 * n)
    for oy in range(10, 24):
        sy, wy = oracle.lanczos_taps(oy, n, 2 * n)
        for ox in range(10, 24):
            sx, wx = oracle.lanczos_taps(ox, n, 2 * n)
            ky, kx = 8 - sy, 8 - sx
            if 0 <= ky < 6 and 0 <= kx < 6:
                v = float(wx[kx]) * float(wy[ky]) / (float(wx.sum()) * float(wy.sum()))
            else:
                v = 0.0
            want = int(np.rint(min(max(v, 0.0), 1.0) * 255.0))
            assert abs(int(out[oy, ox, 0]) - want) <= 1, (ox, oy, out[oy, ox, 0], want)
            assert (out[oy, ox] == out[oy, ox, 0]).all()


def test_s_kat4_border_taps_skipped(oracle):
    """Output column 0 at 2x has start = -3: taps -3,-2,-1 are skipped and the other three are
    renormalised.  A frame that is constant along y isolates the x axis."""
    n = 12
    row = RNG.integers(0, 256, size=(n, 4)).astype(np.uint8)
    f = np.broadcast_to(row[None], (n, n, 4)).copy()
    out = oracle.scale(f, 2 * n, 2 * n)
    s, w = oracle.lanczos_taps(0, n, 2 * n)
    assert s == -3
    wv = w[3:].astype(np.float64)
    want = (wv[:, None] * (row[:3].astype(np.float64) / 255.0)).sum(0) / wv.sum()
    want = np.rint(np.clip(want, 0, 1) * 255.0)
    assert (np.abs(out[n, 0].astype(np.int64) - want) <= 1).all()


def test_scale_roi_matches_full(oracle):
    f = rand_frame(20, 14)
    full = oracle.scale(f, 40, 28)
    part = oracle.scale(f, 40, 28, roi=(5, 3, 33, 20))
    assert (part[3:20, 5:33] == full[3:20, 5:33]).all()
    assert part[:3].sum() == 0 and part[20:].sum() == 0


# ----------------------------------------------------------------- motion.comp

def test_m_kat1_zero_frames(oracle):
    """All-zero frames: every candidate costs 0; strict '<' keeps the first one scanned (F6)."""
    z = np.zeros((12, 20, 4), np.uint8)
    mv = oracle.motion(z, z)
    assert (mv == -16.0).all()


def test_m_kat2_flat_colour(oracle):
    """Flat non-zero colour: (-16,-16) wherever the block shifted by (-16,-16) stays inside prev;
    elsewhere the first candidate in scan order whose shifted block is fully in bounds."""
    H, W = 44, 52
    f = np.empty((H, W, 4), np.uint8)
    f[...] = (40, 90, 200, 255)
    mv = oracle.motion(f, f)
    for (x, y) in [(30, 30), (W - 1, H - 1), (20, 20), (25, 40)]:
        assert tuple(mv[y, x]) == (-16.0, -16.0)
    # pixel (0,0): block is x,y in [-4,3] -> in-bounds part [0,3]; the first dy keeping rows
    # 0..3 inside prev is dy = 0, the first such dx is 0.
    assert tuple(mv[0, 0]) == (0.0, 0.0)
    # pixel (10, 30): block columns 6..13 need dx >= -6; rows 26..33 allow dy = -16.
    assert tuple(mv[30, 10]) == (-6.0, -16.0)


@pytest.mark.parametrize("shift", [(3, -2), (-16, 16), (0, 0), (16, -16), (-7, 5)])
def test_m_kat3_pure_translation(oracle, shift):
    """curr(q) = prev(q - t) on random texture -> MV = -t wherever block and displaced block are
    in bounds (exact zero cost, independent of summation order)."""
    W, H = 72, 64
    prev = synth.make_prev(W, H, seed=77)
    curr = synth.translate(prev, shift, seed=77)
    roi = (24, 24, W - 24, H - 24)
    mv = oracle.motion(prev, curr, roi=roi)
    sub = mv[roi[1]:roi[3], roi[0]:roi[2]]
    assert (sub[..., 0] == -shift[0]).all() and (sub[..., 1] == -shift[1]).all()


def test_motion_small_params(oracle):
    """blockSize / searchRadius are push constants (motion.comp:9-13), not literals."""
    prev = synth.make_prev(24, 20, seed=5)
    curr = synth.translate(prev, (1, -2), seed=5)
    mv = oracle.motion(prev, curr, block_size=4, search_radius=3.0)
    assert (mv[8:12, 8:16, 0] == -1.0).all() and (mv[8:12, 8:16, 1] == 2.0).all()
    assert mv.min() >= -3.0 and mv.max() <= 3.0


# ------------------------------------------------------------ interpolate.comp

def test_i_kat1_zero_mv_is_plain_blend(oracle):
    W, H = 33, 21
    p, c = rand_frame(W, H), rand_frame(W, H)
    mv = np.zeros((H, W, 2), np.float32)
    for t in (0.25, 0.5, 0.75):
        out = oracle.interpolate(p, c, mv, t)
        want = p.astype(np.float64) * (1 - t) + c.astype(np.float64) * t
        assert (np.abs(out.astype(np.float64) - want) <= 0.5 + 1e-3).all()


def test_i_kat2_full_scan_corner_mv_gives_black(oracle):
    """MV = (-16,-16), t = 0.5: both displaced UVs leave [0,1] -> every channel, alpha included, is 0 (F5)."""
    W, H = 40, 24
    p, c = rand_frame(W, H), rand_frame(W, H)
    mv = np.full((H, W, 2), -16.0, np.float32)
    assert (oracle.interpolate(p, c, mv, 0.5) == 0).all()


def test_i_kat3_identical_frames(oracle):
    W, H = 29, 17
    p = rand_frame(W, H)
    mv = np.zeros((H, W, 2), np.float32)
    for t in (0.25, 0.5, 0.75):
        assert (oracle.interpolate(p, p, mv, t) == p).all()


def test_i_literal_pixel_units_in_uv(oracle):
    """MV = (+1, 0), t = 0.5: prev is sampled at uv - 0.5 (half an image to the left, or black),
    curr at uv + 0.5 -- pixel units added to normalised UV, exactly as interpolate.comp:16 does."""
    W, H = 32, 8
    p, c = rand_frame(W, H), rand_frame(W, H)
    mv = np.zeros((H, W, 2), np.float32)
    mv[..., 0] = 1.0
    out = oracle.interpolate(p, c, mv, 0.5).astype(np.float64)
    x = 24                                    # uv.x = 24.5/32; prev sampled at texel 8, curr out of range
    want = p[:, x - W // 2].astype(np.float64) * 0.5
    assert (np.abs(out[:, x] - want) <= 0.5 + 1e-3).all()
    x = 4                                     # prev out of range, curr at texel 20
    want = c[:, x + W // 2].astype(np.float64) * 0.5
    assert (np.abs(out[:, x] - want) <= 0.5 + 1e-3).all()


# ---- opt-in "intended" semantics of the oracle: never the parity default

def test_intended_motion_prefers_the_shortest_vector_among_equal_costs():
    import oracle
    z = np.zeros((24, 40, 4), np.uint8)
    assert (oracle.motion(z, z, semantics=oracle.INTENDED) == 0).all()            # reference: (-16,-16) (M-KAT1)
    f = np.empty((30, 44, 4), np.uint8); f[...] = (9, 99, 199, 255)
    assert (oracle.motion(f, f, semantics=oracle.INTENDED) == 0).all()            # even at the borders: (0,0) costs 0
    rng = np.random.default_rng(3)
    prev = rng.integers(0, 256, size=(40, 56, 4), dtype=np.uint8)
    curr = np.roll(prev, (2, -3), (0, 1))
    a = oracle.motion(prev, curr, roi=(20, 18, 36, 24))[18:24, 20:36]
    b = oracle.motion(prev, curr, roi=(20, 18, 36, 24), semantics=oracle.INTENDED)[18:24, 20:36]
    assert (a == b).all() and (a[..., 0] == 3).all() and (a[..., 1] == -2).all()  # unique minimum: both agree


def test_intended_interpolate_displaces_by_pixels():
    """MV = (dx, dy) pixels: prev is sampled at p - MV*t, curr at p + MV*(1-t) (the shader's own signs); with
    MV*t integral the samples are exact texels."""
    import oracle
    rng = np.random.default_rng(4)
    h, w = 20, 32
    prev = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    curr = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    mv = np.zeros((h, w, 2), np.float32); mv[..., 0] = 4.0; mv[..., 1] = -2.0
    out = oracle.interpolate(prev, curr, mv, 0.5, semantics=oracle.INTENDED)
    y, x = 10, 16
    want = np.round((prev[y + 1, x - 2].astype(np.float64) + curr[y - 1, x + 2]) / 2.0 - 1e-9)   # .5 ties: half-even
    got = out[y, x].astype(np.float64)
    assert (np.abs(got - (prev[y + 1, x - 2].astype(np.float64) + curr[y - 1, x + 2]) / 2.0) <= 0.5).all(), (got, want)
    lit = oracle.interpolate(prev, curr, mv, 0.5)
    assert (lit[y, x] == 0).all()            # reference semantics: both samples leave [0,1] (I-KAT2's mechanism)
