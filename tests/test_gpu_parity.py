"""Parity of the HIP path (through the C-ABI) against the CPU oracle, on a real MI355X.

Bars (BASELINE.json north_star): motion vectors bit-exact; scale and interpolate within +-1 LSB per
8-bit channel (interpolate is in fact held to exact equality: the kernel repeats the oracle's
operation order).  Small cases compare whole frames; full-size cases (1080p -> 4K) use regions of
interest plus size-independent properties, so the oracle finishes in seconds.
"""
import numpy as np
import pytest

from linux_fg_amd import synth

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(2024)


@pytest.fixture(scope="module")
def ctx():
    from linux_fg_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def rand_frame(w, h):
    return RNG.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


def run_scale(ctx, frame, ow, oh):
    src = ctx.frame_from(frame)
    dst = ctx.create_frame(ow, oh)
    ctx.scale(src, dst)
    out = ctx.download(dst)
    ctx.destroy_frame(src); ctx.destroy_frame(dst)
    return out


def run_motion(ctx, prev, curr, bs=8, R=16.0):
    from linux_fg_amd import capi
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    mv = ctx.create_frame(prev.shape[1], prev.shape[0], capi.FORMAT_MV_S8X2)
    ctx.motion(p, c, mv, bs, R)
    out = ctx.download(mv)
    for f in (p, c, mv):
        ctx.destroy_frame(f)
    return out


def run_interpolate(ctx, prev, curr, mv_i8, t):
    from linux_fg_amd import capi
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    m = ctx.frame_from(mv_i8, capi.FORMAT_MV_S8X2)
    o = ctx.create_frame(prev.shape[1], prev.shape[0])
    ctx.interpolate(p, c, m, o, t)
    out = ctx.download(o)
    for f in (p, c, m, o):
        ctx.destroy_frame(f)
    return out


def assert_within_1lsb(got, want, max_mismatch=0.05):
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"max |diff| = {d.max()} at {np.unravel_index(d.argmax(), d.shape)}"
    frac = float((d != 0).mean())
    assert frac <= max_mismatch, f"{frac:.4%} of bytes differ by 1 LSB"
    return frac


# ------------------------------------------------------------------------------ scale

@pytest.mark.parametrize("in_wh,out_wh", [
    ((64, 36), (128, 72)),        # exact 2x -> LDS-tiled kernel (golden fixture shape)
    ((200, 75), (400, 150)),      # 2x, width not a multiple of the 64-column strip
    ((67, 29), (134, 58)),        # 2x, odd sizes, several partial strips
    ((37, 23), (53, 41)),         # generic path
    ((80, 48), (40, 24)),         # down-scale, generic path
    ((33, 17), (33, 17)),         # identity (S-KAT2)
    ((5, 3), (64, 40)),           # tiny input: most taps skipped at the borders
])
def test_scale_matches_oracle(ctx, oracle, in_wh, out_wh):
    f = rand_frame(*in_wh)
    got = run_scale(ctx, f, *out_wh)
    want = oracle.scale(f, *out_wh)
    assert_within_1lsb(got, want)


def test_scale_constant_colour(ctx):
    f = np.empty((40, 64, 4), np.uint8)
    f[...] = (10, 200, 77, 255)
    for (ow, oh) in [(128, 80), (100, 33)]:
        assert (run_scale(ctx, f, ow, oh) == f[0, 0]).all()      # S-KAT1, both kernels


def test_format_load_uscaled_exact(ctx):
    """The 2x kernel takes its texels through buffer_load_format_xyzw with an 8_8_8_8 USCALED descriptor: the texture
    address unit converts byte k to (float)k.  A CONSTANT frame must come back as the same constant (the taps' weights
    sum to 1 within a few ulp; S-KAT1), which only holds for all 256 byte values, in every channel position, if that
    conversion is exact."""
    src = np.empty((16, 256, 4), np.uint8)
    for k in range(256):
        src[:] = (k, 255 - k, k ^ 0x55, (37 * k) & 0xFF)
        got = run_scale(ctx, src, 512, 32)
        assert (got == src[0, 0]).all(), f"byte value {k}"


def test_scale_identity_exact(ctx):
    f = rand_frame(96, 50)
    assert (run_scale(ctx, f, 96, 50) == f).all()                # S-KAT2


def test_scale_config1_540p_to_1080p_full_frame(ctx, oracle):
    """BASELINE config 1 (960x540 -> 1920x1080, Lanczos only): every pixel against the oracle, on the benchmark's
    synthetic frame and on uniform noise (SURVEY.md section 7 step 3 names this shape as the first parity case)."""
    prev, _ = synth.make_pair(960, 540, stream=0)
    for src in (prev, synth.noise_bytes(960, 540, 540960)):
        got = run_scale(ctx, src, 1920, 1080)
        assert_within_1lsb(got, oracle.scale(src, 1920, 1080))


def test_scale_1080p_to_4k_roi(ctx, oracle):
    """BASELINE config 2 at full size: ROI spot checks against the oracle plus a 2x vs generic-kernel
    cross-check would need a second entry point; the ROIs cover corners, edges and the interior."""
    prev, _ = synth.make_pair(1920, 1080, stream=0)
    got = run_scale(ctx, prev, 3840, 2160)
    for roi in [(0, 0, 96, 40), (3744, 2120, 3840, 2160), (1800, 1000, 1960, 1040), (0, 2100, 64, 2160),
                (3700, 0, 3840, 24)]:
        want = oracle.scale(prev, 3840, 2160, roi=roi)
        x0, y0, x1, y1 = roi
        assert_within_1lsb(got[y0:y1, x0:x1], want[y0:y1, x0:x1])


def test_scale_1080p_to_4k_every_pixel_repeated(ctx, oracle):
    """Every pixel of the full-size output, on several runs and both frames of the benchmark pair: catches the rare,
    timing-dependent faults a region check misses (a store-data hazard once left a few dozen pixels per frame holding
    an unpacked float, on some runs only).  Also the property the motion stage leans on: an exact 2x upscale commutes
    with integer translations away from the rim (up to the last bit: the filter weights come from per-column float
    arithmetic)."""
    prev, curr = synth.make_pair(1920, 1080, stream=0, shift=(3, -2))
    ups = []
    for src in (prev, curr):
        want = oracle.scale(src, 3840, 2160)
        s, d = ctx.frame_from(src), ctx.create_frame(3840, 2160)
        for _ in range(4):
            ctx.scale(s, d)
            ctx.sync()
            got = ctx.download(d)
            assert_within_1lsb(got, want)
        ups.append(got)
        ctx.destroy_frame(s)
        ctx.destroy_frame(d)
    P, C = ups
    d = np.abs(C[16:2160 - 16, 16:3840 - 16].astype(np.int16) - P[16 + 4:2160 - 16 + 4, 16 - 6:3840 - 16 - 6].astype(np.int16))
    assert d.max() <= 1 and float((d != 0).mean()) < 1e-3


# ------------------------------------------------------------------------------ motion

def as_int(mv_f32):
    return mv_f32.astype(np.int8)


@pytest.mark.parametrize("wh,shift", [((128, 64), (3, -2)), ((70, 45), (-5, 7)), ((64, 32), (0, 0)),
                                      ((200, 40), (16, -16)), ((33, 90), (-16, 16)),
                                      # purely vertical pans: the strip they expose is a few rows over the whole width -- the row
                                      # band of the prefilter (prefilter_rowband.inc, "Row band"), at the bottom and at the top, in a segment's
                                      # upper and lower half (160 rows: the last segment starts at row 144)
                                      ((224, 160), (0, 6)), ((224, 160), (0, -6)), ((180, 150), (0, -3)), ((180, 150), (1, 5))])
def test_motion_translation_matches_oracle(ctx, oracle, wh, shift):
    prev = synth.make_prev(*wh, seed=11)
    curr = synth.translate(prev, shift, seed=11)
    got = run_motion(ctx, prev, curr)
    want = as_int(oracle.motion(prev, curr))
    assert (got == want).all(), f"{(got != want).any(-1).sum()} pixels differ"


def test_motion_uncorrelated_matches_oracle(ctx, oracle):
    """No good match anywhere: minima are decided by the last bits of the 64-term fp32 chains."""
    prev, curr = synth.make_uncorrelated_pair(96, 64, stream=3)
    got = run_motion(ctx, prev, curr)
    want = as_int(oracle.motion(prev, curr))
    assert (got == want).all(), f"{(got != want).any(-1).sum()} pixels differ"


def test_motion_smooth_content_matches_oracle(ctx, oracle):
    """Low-contrast smooth content: many near-ties between candidates."""
    y, x = np.mgrid[0:48, 0:80]
    base = ((x * 3 + y * 2) // 4).astype(np.uint8)
    prev = np.stack([base, base + 1, 255 - base, np.full_like(base, 255)], -1)
    curr = np.roll(prev, (1, 2), (0, 1))
    got = run_motion(ctx, prev, curr)
    want = as_int(oracle.motion(prev, curr))
    assert (got == want).all(), f"{(got != want).any(-1).sum()} pixels differ"


def test_motion_kat_zero_and_flat(ctx, oracle):
    z = np.zeros((40, 72, 4), np.uint8)
    assert (run_motion(ctx, z, z) == -16).all()                  # M-KAT1 (F6)
    f = np.empty((44, 52, 4), np.uint8)
    f[...] = (40, 90, 200, 255)
    got = run_motion(ctx, f, f)
    assert (got == as_int(oracle.motion(f, f))).all()            # M-KAT2, edges included
    assert tuple(got[30, 30]) == (-16, -16) and tuple(got[0, 0]) == (0, 0) and tuple(got[30, 10]) == (-6, -16)


@pytest.mark.parametrize("bs,R", [(4, 3.0), (8, 2.0), (5, 4.0), (16, 1.0), (8, 0.0)])
def test_motion_other_parameters_match_oracle(ctx, oracle, bs, R):
    """blockSize / searchRadius are push constants in the reference (motion.comp:9-13): generic kernel."""
    prev = synth.make_prev(50, 34, seed=21)
    curr = synth.translate(prev, (1, -1), seed=21)
    got = run_motion(ctx, prev, curr, bs, R)
    want = as_int(oracle.motion(prev, curr, bs, R))
    assert (got == want).all()


def test_motion_4k_translation_property(ctx, oracle):
    """BASELINE config 3 size.  M-KAT3: curr(q) = prev(q - t) -> MV = -t wherever the block and the
    displaced block are in bounds; plus ROI checks against the oracle at the borders."""
    W, H = 3840, 2160
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED)
    curr = synth.translate(prev, (3, -2), synth.BASE_SEED)
    got = run_motion(ctx, prev, curr)
    inner = got[24:H - 24, 24:W - 24]
    assert (inner[..., 0] == -3).all() and (inner[..., 1] == 2).all()
    for roi in [(0, 0, 40, 12), (W - 40, H - 12, W, H), (0, H - 10, 48, H), (W - 36, 0, W, 10), (1900, 1070, 1932, 1078)]:
        want = as_int(oracle.motion(prev, curr, roi=roi))
        x0, y0, x1, y1 = roi
        assert (got[y0:y1, x0:x1] == want[y0:y1, x0:x1]).all(), roi


def run_motion_mode(ctx, prev, curr, mode):
    from linux_fg_amd import capi
    ctx.set_motion_mode(mode)
    try:
        out = run_motion(ctx, prev, curr)
        stats = ctx.motion_last_stats() if mode == capi.MOTION_PREFILTERED else None
    finally:
        ctx.set_motion_mode(capi.MOTION_PREFILTERED)
    return out, stats


def _adversarial_pairs():
    """Frame pairs aimed at the prefilter's error bracket (csrc/motion_prefilter.hip, "Bracket") and its bookkeeping."""
    rng = np.random.default_rng(77)
    w, h = 200, 150                                    # 4 x 3 prefilter tiles, all partial at the far edges
    hi = rng.integers(254, 256, size=(h, w, 4), dtype=np.uint8)          # byte differences of 1 at the top of the
    hi2 = rng.integers(254, 256, size=(h, w, 4), dtype=np.uint8)         # range: largest fp32 error of a texel difference
    lo = rng.integers(0, 2, size=(h, w, 4), dtype=np.uint8)
    lo2 = rng.integers(0, 2, size=(h, w, 4), dtype=np.uint8)
    noise = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    noise2 = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    # periodic texture: many candidates with almost the same cost (period 4 in x, 6 in y) plus faint noise
    yy, xx = np.mgrid[0:h, 0:w]
    tex = ((xx % 4) * 40 + (yy % 6) * 20).astype(np.uint8)
    per = np.stack([tex, tex // 2, 255 - tex, np.full_like(tex, 255)], -1)
    per2 = per.copy()
    per2[..., 0] ^= rng.integers(0, 2, size=(h, w), dtype=np.uint8)
    # half flat with a brightness change (every candidate ties at a NON-zero cost: list overflow -> exact
    # fallback), half noise
    mix = noise.copy(); mix[:, : w // 2] = (10, 20, 30, 255)
    mix2 = noise2.copy(); mix2[:, : w // 2] = (12, 20, 30, 255)
    # static frame with flat areas: many zero-cost candidates per pixel, which must NOT need the fallback
    stat = noise.copy(); stat[40:110, 30:150] = (90, 90, 90, 255); stat[:, 160:] = 0
    return {"high bytes": (hi, hi2), "low bytes": (lo, lo2), "noise": (noise, noise2),
            "periodic": (per, per2), "flat + noise": (mix, mix2), "static with flat areas": (stat, stat.copy())}


@pytest.mark.parametrize("name", ["high bytes", "low bytes", "noise", "periodic", "flat + noise", "static with flat areas"])
def test_motion_prefilter_bracket_adversarial(ctx, oracle, name):
    """The prefiltered path must return the oracle's MVs on content built to stress its bracket: unit byte
    differences at both ends of the range, uncorrelated noise, near-tied periodic texture, flat areas tied at
    a non-zero cost that overflow the candidate lists (those tiles must come back through the exact kernel), and a
    static frame whose flat areas give many zero-cost candidates (handled without lists or fallback)."""
    from linux_fg_amd import capi
    prev, curr = _adversarial_pairs()[name]
    want = as_int(oracle.motion(prev, curr))
    got, stats = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    assert (got == want).all(), f"{(got != want).any(-1).sum()} pixels differ ({name}, stats {stats})"
    exact, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (exact == want).all()
    if name == "flat + noise":
        assert stats[1] > 0, "flat half should have fallen back to the exact kernel"
    if name in ("noise", "static with flat areas"):       # (zero motion is visited first: static rims stay cheap)
        assert stats[1] == 0, f"{name} must not overflow the lists"


def test_motion_prefilter_zero_cost_candidates_need_no_fallback(ctx, oracle):
    """Static frames: a flat one (all 1089 candidates cost exactly 0 in the interior) and a noisy one (only m = 0
    does).  The zero-cost rule keeps both off the exact kernel and still returns the oracle's first-in-scan-order
    answer."""
    from linux_fg_amd import capi
    flat = np.empty((130, 120, 4), np.uint8); flat[...] = (33, 66, 99, 255)
    noisy = np.random.default_rng(5).integers(0, 256, size=(130, 120, 4), dtype=np.uint8)
    for frame in (flat, noisy):
        got, stats = run_motion_mode(ctx, frame, frame.copy(), capi.MOTION_PREFILTERED)
        assert (got == as_int(oracle.motion(frame, frame))).all()
        assert stats[1] == 0, stats


@pytest.mark.parametrize("wh", [(56, 64), (57, 65), (113, 129), (300, 70), (64, 200)])
def test_motion_prefilter_equals_exact_kernel_on_ragged_sizes(ctx, wh):
    """Prefilter tiles are 56 x 64, exact tiles 64 x 64: sizes around their multiples, both modes identical."""
    from linux_fg_amd import capi
    w, h = wh
    prev = synth.make_prev(w, h, seed=w * 1000 + h)
    curr = synth.translate(prev, (-4, 3), seed=w * 1000 + h)
    a, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all()


def test_motion_4k_modes_agree_on_noise_and_translation(ctx):
    """BASELINE config 3 size, every tile and every shared-tile work unit: the prefiltered path against the
    literal kernel on uncorrelated noise (nothing ties, ~7 records per pixel) and on the translated pair."""
    from linux_fg_amd import capi
    W, H = 3840, 2160
    noise_a = synth.noise_bytes(W, H, 111)
    noise_b = synth.noise_bytes(W, H, 222)
    a, stats = run_motion_mode(ctx, noise_a, noise_b, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, noise_a, noise_b, capi.MOTION_EXACT_ONLY)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ"
    # (records HELD at the end, not recorded over the search -- 7.5: a pixel's count restarts whenever a candidate undercuts
    #  its threshold by more than the bracket's width, which on noise is nearly every new minimum)
    assert stats[1] == 0 and 1.0 <= stats[2] < 3.0, stats
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED)
    curr = synth.translate(prev, (-7, 11), synth.BASE_SEED)
    a, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all()


def test_motion_modes_agree_on_benchmark_frames_and_occlusions(ctx):
    """The benchmark's own frames (a translated 1080p pair, both upscaled on the device: small non-zero costs at the
    true vector, exposed bands at the rim) and the same pair with patches of fresh noise pasted into curr (segments in
    the middle of the frame that find no match and search in full next to segments that close at once): prefiltered
    path == literal kernel, nothing through the fallback."""
    from linux_fg_amd import capi
    W, H = 3840, 2160
    pin = synth.make_prev(W // 2, H // 2, seed=synth.BASE_SEED)
    cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
    P = run_scale(ctx, pin, W, H)
    C = run_scale(ctx, cin, W, H)
    a, stats = run_motion_mode(ctx, P, C, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, P, C, capi.MOTION_EXACT_ONLY)
    assert (a == b).all() and stats[1] == 0
    inner = a[40:H - 40, 40:W - 40]
    assert (inner[..., 0] == -6).all() and (inner[..., 1] == 4).all()
    C2 = C.copy()
    patch = synth.noise_bytes(W, H, 4242)
    for (x0, y0, w, h) in [(500, 300, 90, 70), (1900, 1000, 200, 33), (3000, 1700, 17, 150), (1234, 2000, 300, 100)]:
        C2[y0:y0 + h, x0:x0 + w] = patch[y0:y0 + h, x0:x0 + w]
    a, stats = run_motion_mode(ctx, P, C2, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, P, C2, capi.MOTION_EXACT_ONLY)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ"
    assert stats[1] == 0


@pytest.mark.parametrize("amp", [1, 2, 6])
def test_motion_modes_agree_under_sensor_noise(ctx, amp):
    """A translated frame with +-amp levels of independent noise in every channel: matches cost a few hundred instead
    of zero, the one-point partial-distortion test rarely fires and the four-point sums (interior and rim variants)
    carry the search.  Prefiltered path == literal kernel; the translation is still found away from the rim."""
    from linux_fg_amd import capi
    W, H = 1920, 1080
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED + 21)
    moved = synth.translate(prev, (-4, 7), synth.BASE_SEED + 21)
    n = synth.noise_bytes(W, H, 31337 + amp) % (2 * amp + 1)
    curr = np.clip(moved.astype(np.int16) + n.astype(np.int16) - amp, 0, 255).astype(np.uint8)
    a, stats = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ"
    assert stats[1] == 0
    inner = a[40:H - 40, 40:W - 40]
    assert (inner[..., 0] == 4).mean() > 0.999 and (inner[..., 1] == -7).mean() > 0.999


def test_motion_flat_areas_with_a_brightness_change(ctx, oracle):
    """Flat frames one grey level apart: every candidate ties at a non-zero cost.  Where the whole search window is
    one colour the prefilter knows the answer (the first candidate in tie order) without a search; tiles at the rim
    (zero fill in the window) and at the edge of a textured inset still go through the exact kernel.  Both paths
    agree everywhere, and most tiles stay out of the fallback."""
    from linux_fg_amd import capi
    W, H = 1920, 1080
    prev = np.full((H, W, 4), 100, np.uint8)
    curr = np.full((H, W, 4), 101, np.uint8)
    tex = synth.make_prev(W, H, seed=synth.BASE_SEED + 77)
    prev[400:700, 800:1200] = tex[400:700, 800:1200]
    curr[400:700, 800:1200] = tex[398:698, 803:1203]
    a, stats = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ"
    assert stats[1] < stats[0] // 2, stats                     # the flat interior stayed out of the exact kernel
    assert (a[100, 300] == (-16, -16)).all()                  # all candidates tie: the first in scan order (F6)
    small = (slice(0, 80), slice(0, 120))
    want = as_int(oracle.motion(prev[small], curr[small]))
    got = run_motion(ctx, prev[small].copy(), curr[small].copy())
    assert (got == want).all()


def _mixed_pair(w, h, seed, max_amp=4):
    """A frame pair that mixes what the prefilter treats differently: a translation, sensor noise on part of the
    frame, patches of fresh noise, a static flat area and a flat area one level apart (ties at a non-zero cost)."""
    rng = np.random.default_rng(seed)
    prev = synth.make_prev(w, h, seed=seed)
    shift = (int(rng.integers(-12, 13)), int(rng.integers(-12, 13)))
    curr = synth.translate(prev, shift, seed)
    if rng.random() < 0.7:                                   # sensor noise on a band of rows
        y0, y1 = sorted(int(v) for v in rng.integers(0, h, 2))
        amp = int(rng.integers(1, max_amp + 1))     # (max_amp 4: the suite's seeded cases; tools/fuzz_motion_4k.py asks for more)
        n = synth.noise_bytes(w, h, seed + 1) % (2 * amp + 1)
        noisy = np.clip(curr.astype(np.int16) + n.astype(np.int16) - amp, 0, 255).astype(np.uint8)
        curr[y0:y1] = noisy[y0:y1]
    fresh = synth.noise_bytes(w, h, seed + 2)
    for _ in range(int(rng.integers(0, 4))):                 # occlusions
        pw, ph = int(rng.integers(4, max(5, w // 6))), int(rng.integers(4, max(5, h // 6)))
        x0, y0 = int(rng.integers(0, w - pw)), int(rng.integers(0, h - ph))
        curr[y0:y0 + ph, x0:x0 + pw] = fresh[y0:y0 + ph, x0:x0 + pw]
    if rng.random() < 0.5:                                   # static flat area (zero-cost ties)
        pw, ph = int(rng.integers(8, max(9, w // 4))), int(rng.integers(8, max(9, h // 4)))
        x0, y0 = int(rng.integers(0, w - pw)), int(rng.integers(0, h - ph))
        prev[y0:y0 + ph, x0:x0 + pw] = 90
        curr[y0:y0 + ph, x0:x0 + pw] = 90
    if rng.random() < 0.3:                                   # a small fade (non-zero ties: overflow, exact-kernel tiles)
        pw, ph = int(rng.integers(20, 60)), int(rng.integers(20, 60))
        x0, y0 = int(rng.integers(0, max(1, w - pw))), int(rng.integers(0, max(1, h - ph)))
        prev[y0:y0 + ph, x0:x0 + pw] = 50
        curr[y0:y0 + ph, x0:x0 + pw] = 51
    return prev, curr


def _full_hd_mixture():
    """1920x1080 pair with everything the prefilter treats differently in one frame: a translation, a band of
    sensor noise, patches of fresh noise (segments without a match, hand-over), patches that move on their own,
    a static flat area (zero-cost ties), a flat area one grey level apart (non-zero ties: list overflow, exact-kernel
    tiles), and one of each touching the image border (rim segment units, plateaus)."""
    W, H = 1920, 1080
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED + 1080)
    curr = synth.translate(prev, (-5, 9), synth.BASE_SEED + 1080)
    n = synth.noise_bytes(W, H, 271828) % 5
    noisy = np.clip(curr.astype(np.int16) + n.astype(np.int16) - 2, 0, 255).astype(np.uint8)
    curr[300:520] = noisy[300:520]
    fresh = synth.noise_bytes(W, H, 314159)
    for (x0, y0, w, h) in [(200, 100, 150, 90), (1000, 640, 64, 200), (1700, 900, 220, 180), (0, 700, 40, 120)]:
        curr[y0:y0 + h, x0:x0 + w] = fresh[y0:y0 + h, x0:x0 + w]
    for (x0, y0, w, h, dx, dy) in [(600, 150, 120, 80, 7, -3), (1300, 400, 90, 140, -11, 6), (900, 950, 200, 60, 2, 12)]:
        curr[y0:y0 + h, x0:x0 + w] = prev[y0 - dy:y0 - dy + h, x0 - dx:x0 - dx + w]
    prev[560:680, 100:420] = 37; curr[560:680, 100:420] = 37                 # static flat area
    prev[760:840, 1200:1330] = 50; curr[760:840, 1200:1330] = 51             # fade patch in the interior
    prev[0:70, 1500:1640] = 200; curr[0:70, 1500:1640] = 201                 # fade patch on the top border
    return prev, curr


def test_motion_full_hd_both_modes_against_the_oracle(ctx, oracle):
    """One full-frame ORACLE comparison at 1920x1080 (about a minute of CPU): whole-tile work units, rim segment
    units, the hand-over queue, both partial-distortion tests, plateaus, the shared-tile merge and the exact-kernel
    fallback are all checked against the oracle itself here, for both motion modes -- everywhere else frames of this
    size are only compared HIP against HIP."""
    from linux_fg_amd import capi
    prev, curr = _full_hd_mixture()
    want = as_int(oracle.motion(prev, curr))
    a, stats = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    assert (a == want).all(), f"prefiltered: {(a != want).any(-1).sum()} pixels differ from the oracle"
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (b == want).all(), f"exact only: {(b != want).any(-1).sum()} pixels differ from the oracle"
    assert 0 < stats[1] < stats[0] // 4, stats                 # the fade patches did go through the fallback
    inner = want[540:556, 700:1100]                            # clean translation below the noise band
    assert (inner[..., 0] == 5).all() and (inner[..., 1] == -9).all()


def _uhd_mixture():
    """3840x2160 pair of the benchmark's kind (a 1080p pan, both frames upscaled 2x by the device's own Lanczos kernel would be
    the real thing; here the frames are synthesised at 4K directly so that the oracle and the kernels see the same bytes
    without a scale stage in between) with everything the sweep's contents hold, each in a known place: a pan, a band of
    sensor noise, patches of fresh noise (hand-over, full search), patches that move on their own, a static flat area and a
    flat area one grey level apart (fallback tiles), and one of each at an image border."""
    W, H = 3840, 2160
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED + 2160)
    curr = synth.translate(prev, (6, -4), synth.BASE_SEED + 2160)
    n = synth.noise_bytes(W, H, 161803) % 5
    noisy = np.clip(curr.astype(np.int16) + n.astype(np.int16) - 2, 0, 255).astype(np.uint8)
    curr[900:1100] = noisy[900:1100]
    fresh = synth.noise_bytes(W, H, 577215)
    for (x0, y0, w, h) in [(400, 300, 260, 150), (2500, 1500, 96, 300), (3700, 1900, 140, 260), (0, 1300, 50, 200)]:
        curr[y0:y0 + h, x0:x0 + w] = fresh[y0:y0 + h, x0:x0 + w]
    for (x0, y0, w, h, dx, dy) in [(1200, 200, 300, 160, 9, -5), (2900, 600, 180, 260, -13, 7), (1700, 1800, 400, 120, 3, 14)]:
        curr[y0:y0 + h, x0:x0 + w] = prev[y0 - dy:y0 - dy + h, x0 - dx:x0 - dx + w]
    prev[1250:1420, 300:800] = 37; curr[1250:1420, 300:800] = 37                 # static flat area
    prev[1600:1690, 2200:2340] = 50; curr[1600:1690, 2200:2340] = 51             # fade patch in the interior
    prev[0:80, 3000:3150] = 200; curr[0:80, 3000:3150] = 201                     # fade patch on the top border
    return prev, curr


def test_motion_4k_mixture_regions_against_the_oracle(ctx, oracle):
    """BASELINE config 3's size against the ORACLE itself, on a frame that mixes everything the prefilter treats differently
    (round 2 compared such frames at this size HIP against HIP only, and the oracle saw 4K through border regions of a pure
    pan).  The oracle is too slow for all 8.3 M pixels, so it walks regions placed where each mechanism is at work: the
    four borders and a corner (rim segment units, plateaus), the edge and the inside of an occlusion (hand-over, narrow
    search, full search), the edge of a patch that moves on its own, the noise band's boundary (four- and sixteen-point
    tests), the static flat area's rim (zero-cost ties), both fade patches (fallback tiles, shared and merged) -- and both
    motion modes have to agree with it bit for bit there, and with each other everywhere."""
    from linux_fg_amd import capi
    prev, curr = _uhd_mixture()
    a, stats = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ between the two modes"
    assert 0 < stats[1] < 64, stats                                # the fade patches went through the fallback, little else
    W, H = 3840, 2160
    rois = [(0, 0, 96, 40), (W - 96, 0, W, 40), (0, H - 40, 96, H), (W - 96, H - 40, W, H),      # corners
            (1800, 0, 1960, 24), (1800, H - 24, 1960, H), (0, 1000, 24, 1100), (W - 24, 1000, W, 1100),   # borders (noise band on two)
            (380, 280, 460, 330), (500, 350, 560, 380),            # an occlusion: its corner, its inside
            (2480, 1480, 2520, 1560),                              # the narrow occlusion's edge
            (1180, 180, 1260, 230), (2880, 840, 2960, 880),        # patches that move on their own: corners
            (2000, 880, 2100, 920), (2000, 1080, 2100, 1120),      # the noise band's two boundaries
            (280, 1230, 360, 1270), (760, 1400, 840, 1440),        # the static flat area's rim
            (2180, 1580, 2260, 1620), (2980, 60, 3060, 100),       # the fade patches' rims (fallback tiles)
            (0, 1290, 70, 1330)]                                   # the occlusion on the left border
    for roi in rois:
        x0, y0, x1, y1 = roi
        want = as_int(oracle.motion(prev, curr, roi=roi))[y0:y1, x0:x1]
        assert (a[y0:y1, x0:x1] == want).all(), f"prefiltered path differs from the oracle in {roi}: {(a[y0:y1, x0:x1] != want).any(-1).sum()} pixels"
    inner = a[1450:1550, 1000:1600]                                # clean pan between the mechanisms
    assert (inner[..., 0] == -6).all() and (inner[..., 1] == 4).all()


@pytest.mark.parametrize("case", range(60))
def test_motion_modes_agree_on_random_mixtures(ctx, case):
    """Seeded fuzz over frame sizes and content mixtures: small frames (tiles shared between workgroups), mid-size
    and 1080p-class frames (whole tiles, rim segment units, hand-over, both partial-distortion tests, plateaus,
    overflowing tiles) -- the prefiltered path must equal the literal kernel bit for bit."""
    from linux_fg_amd import capi
    rng = np.random.default_rng(9000 + case)
    if case % 3 == 0:
        w, h = int(rng.integers(64, 400)), int(rng.integers(64, 300))
    elif case % 3 == 1:
        w, h = int(rng.integers(400, 1300)), int(rng.integers(300, 900))
    else:
        w, h = int(rng.integers(1300, 2100)), int(rng.integers(1000, 1300))
    prev, curr = _mixed_pair(w, h, 9000 + case)
    if case % 5 == 4:                                        # every fifth case under the opt-in tie order
        ctx.set_semantics(capi.SEMANTICS_INTENDED)
    try:
        a, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
        b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    finally:
        ctx.set_semantics(capi.SEMANTICS_REFERENCE)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ at {w}x{h}"


def test_motion_hand_over_queue_overflows_gracefully(ctx):
    """Segments that find no match are handed to a second launch through a queue with room for a quarter of the
    frame's segments, unless the hint samples say that most of the frame is unmatched.  Here the hint samples see a
    clean translation (islands around the 16 x 16 sample points are left intact) while everything else in curr is fresh
    noise: nearly every segment asks to be handed over, the queue fills up, the rest search in place -- and the vectors
    still equal the literal kernel's."""
    from linux_fg_amd import capi
    W, H = 3840, 2160
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED + 9)
    curr = synth.noise_bytes(W, H, 777)
    moved = synth.translate(prev, (5, 3), synth.BASE_SEED + 9)
    for gy in range(16):
        for gx in range(16):
            cx, cy = (2 * gx + 1) * W // 32, (2 * gy + 1) * H // 32
            curr[cy - 14:cy + 14, cx - 14:cx + 14] = moved[cy - 14:cy + 14, cx - 14:cx + 14]
    a, stats = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ"
    assert stats[1] == 0
    cy, cx = H // 32, W // 32
    assert tuple(a[cy, cx]) == (-5, -3)                     # inside an island the translation is found


def test_handed_over_segments_are_listed_once_then_flagged_tiles_merge_cleanly(ctx):
    """Round 3 listed a handed-over segment twice for the resolve kernel (once by the wave that pushed it, once by the queue
    unit that ran its first part): on a pair where the queue fills up the list of open segments ran past its tiles x 4 words
    into the scratch where the parts of a flagged tile meet, and the NEXT call of that size merged against garbage.  Same
    context, same size, no re-plan in between: a pair that hands over as many segments as the queue holds, then a pair with
    flagged tiles (fade patches) -- the prefiltered path must equal the literal kernel on both, and no segment may be listed
    twice."""
    from linux_fg_amd import capi
    W, H = 3840, 2160
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED + 9)
    curr = synth.noise_bytes(W, H, 778)
    moved = synth.translate(prev, (5, 3), synth.BASE_SEED + 9)
    for gy in range(16):
        for gx in range(16):
            cx, cy = (2 * gx + 1) * W // 32, (2 * gy + 1) * H // 32
            curr[cy - 14:cy + 14, cx - 14:cx + 14] = moved[cy - 14:cy + 14, cx - 14:cx + 14]
    mix_prev, mix_curr = _uhd_mixture()
    ctx.set_motion_mode(capi.MOTION_PREFILTERED)
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    mp, mc = ctx.frame_from(mix_prev), ctx.frame_from(mix_curr)
    mv = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
    try:
        ctx.motion(p, c, mv)
        a1 = ctx.download(mv)
        open1, segments = ctx.motion_open_segments()
        assert segments == ((W + 55) // 56) * ((H + 63) // 64) * 4
        assert 0 < open1 <= segments, (open1, segments)
        ctx.motion(mp, mc, mv)                                   # flagged tiles: their parts meet in the scratch behind the list
        a2 = ctx.download(mv)
        open2, _ = ctx.motion_open_segments()
        assert open2 <= segments, (open2, segments)
        assert ctx.motion_last_stats()[1] > 0                    # (the fade patches did go through the fallback)
        ctx.set_motion_mode(capi.MOTION_EXACT_ONLY)
        ctx.motion(p, c, mv)
        b1 = ctx.download(mv)
        ctx.motion(mp, mc, mv)
        b2 = ctx.download(mv)
    finally:
        ctx.set_motion_mode(capi.MOTION_PREFILTERED)
        for f in (p, c, mp, mc, mv):
            ctx.destroy_frame(f)
    assert (a1 == b1).all(), f"{(a1 != b1).any(-1).sum()} pixels differ on the hand-over pair"
    assert (a2 == b2).all(), f"{(a2 != b2).any(-1).sum()} pixels differ on the pair that followed it"


def test_three_stages_at_8k(ctx, oracle):
    """BASELINE config 5 size (4K -> 8K, three interpolation factors): scale against the oracle on regions, motion by
    the translation property and against the literal kernel everywhere, interpolate exact on regions for each factor."""
    from linux_fg_amd import capi
    w, h, W, H = 3840, 2160, 7680, 4320
    pin = synth.make_prev(w, h, seed=synth.BASE_SEED + 5)
    cin = synth.translate(pin, (-2, 5), synth.BASE_SEED + 5)
    p, c = ctx.frame_from(pin), ctx.frame_from(cin)
    P, C, O = ctx.create_frame(W, H), ctx.create_frame(W, H), ctx.create_frame(W, H)
    M = ctx.create_frame(W, H, capi.FORMAT_MV_S8X2)
    ctx.scale(p, P)
    ctx.scale(c, C)
    ctx.motion(P, C, M)
    ctx.sync()
    Pn, Cn, Mn = ctx.download(P), ctx.download(C), ctx.download(M)
    rois = [(0, 0, 96, 24), (W - 96, H - 24, W, H), (3800, 2150, 3900, 2170)]
    for roi in rois:
        x0, y0, x1, y1 = roi
        assert_within_1lsb(Cn[y0:y1, x0:x1], oracle.scale(cin, W, H, roi=roi)[y0:y1, x0:x1], max_mismatch=0.08)
    inner = Mn[48:H - 48, 48:W - 48]
    assert (inner[..., 0] == 4).all() and (inner[..., 1] == -10).all()
    # ... and against the ORACLE itself where the oracle can afford it at this size (round 3 checked 8K vectors by the
    # translation property and against the literal kernel only): a corner that holds both strips the pan exposes (the top rows
    # and the right columns), the right border's strip alone, the bottom-left corner, and the interior
    for roi in [(W - 72, 0, W, 28), (W - 40, 2000, W, 2032), (0, H - 28, 64, H), (3800, 2150, 3864, 2174), (0, 0, 64, 24)]:
        x0, y0, x1, y1 = roi
        want = as_int(oracle.motion(Pn, Cn, roi=roi))[y0:y1, x0:x1]
        assert (Mn[y0:y1, x0:x1] == want).all(), f"8K vectors differ from the oracle in {roi}: {(Mn[y0:y1, x0:x1] != want).any(-1).sum()} pixels"
    ctx.set_motion_mode(capi.MOTION_EXACT_ONLY)
    try:
        ctx.motion(P, C, M)
        ctx.sync()
        assert (ctx.download(M) == Mn).all()
    finally:
        ctx.set_motion_mode(capi.MOTION_PREFILTERED)
    for t in (0.25, 0.5, 0.75):
        ctx.interpolate(P, C, M, O, t)
        ctx.sync()
        On = ctx.download(O)
        for roi in rois:
            x0, y0, x1, y1 = roi
            want = oracle.interpolate(Pn, Cn, Mn, t, roi=roi)
            assert (On[y0:y1, x0:x1] == want[y0:y1, x0:x1]).all(), (t, roi)
    for f in (p, c, P, C, O, M):
        ctx.destroy_frame(f)


def test_motion_zoom_and_rotation_fields(ctx):
    """Smoothly varying motion (a 4 % zoom, a 1.5 degree rotation of a noise texture): the per-call hints differ from
    block to block, the lists must not overflow, and both modes must agree."""
    from linux_fg_amd import capi
    w, h = 640, 384
    rng = np.random.default_rng(123)
    tex = rng.integers(0, 256, size=(h + 64, w + 64, 4), dtype=np.uint8)
    prev = tex[32:32 + h, 32:32 + w].copy()
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = w / 2.0, h / 2.0
    fields = {"zoom": ((xx - cx) / 1.04 + cx, (yy - cy) / 1.04 + cy)}
    a = np.deg2rad(1.5)
    fields["rotation"] = ((xx - cx) * np.cos(a) - (yy - cy) * np.sin(a) + cx, (xx - cx) * np.sin(a) + (yy - cy) * np.cos(a) + cy)
    for name, (sx, sy) in fields.items():
        ix = np.clip(np.rint(sx).astype(np.int64) + 32, 0, w + 63)
        iy = np.clip(np.rint(sy).astype(np.int64) + 32, 0, h + 63)
        curr = tex[iy, ix]
        got, stats = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
        want, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
        assert (got == want).all(), name
        assert stats[1] == 0, (name, stats)


def test_exact_sqrt_exhaustive(ctx):
    """csrc/lfg_motion_common.hpp: exact_sqrt (one Newton step on v_rsq_f32) against the compiler's IEEE sqrtf for
    every float from 2^-21 to 8 (the motion kernel feeds it sums of four squares in [0, 4], the
    smallest non-zero one being (1/255)^2 ~ 1.5e-5) and for 0."""
    lo = int(np.float32(2.0 ** -21).view(np.uint32))
    hi = int(np.float32(8.0).view(np.uint32))
    assert ctx.selftest_sqrt(lo, hi) == 0
    assert ctx.selftest_sqrt(0, 0) == 0


def test_device_sqrt_and_unorm_match_host(ctx, oracle):
    """The compiler's device sqrtf / the 1/255 form against the host's: a frame pair whose distances
    cover many distinct sums of squares, through the generic (literal) kernel with a 1x1 block and
    zero radius -- the cost is a single distance, so equal MVs say nothing; instead use two
    candidates' ordering: covered by the full parity tests above.  Here: every byte pair once."""
    a = np.arange(256, dtype=np.uint8)
    prev = np.zeros((16, 256, 4), np.uint8)
    curr = np.zeros((16, 256, 4), np.uint8)
    for r in range(16):
        prev[r, :, 0] = a; prev[r, :, 1] = np.roll(a, 17 * r); prev[r, :, 2] = a[::-1]; prev[r, :, 3] = np.roll(a, 5 * r + 1)
        curr[r, :, 0] = np.roll(a, 3 * r + 1); curr[r, :, 1] = a; curr[r, :, 2] = np.roll(a, 7 * r); curr[r, :, 3] = a[::-1]
    got = run_motion(ctx, prev, curr, 3, 5.0)
    want = as_int(oracle.motion(prev, curr, 3, 5.0))
    assert (got == want).all()


# ------------------------------------------------------------------------------ interpolate

@pytest.mark.parametrize("t", [0.25, 0.5, 0.75, 0.3])
def test_interpolate_random_mv_exact(ctx, oracle, t):
    W, H = 96, 40
    p, c = rand_frame(W, H), rand_frame(W, H)
    mv = RNG.integers(-2, 3, size=(H, W, 2)).astype(np.int8)
    mv[RNG.random((H, W)) < 0.5] = 0
    got = run_interpolate(ctx, p, c, mv, t)
    want = oracle.interpolate(p, c, mv.astype(np.float32), t)
    assert (got == want).all(), f"{(got != want).sum()} bytes differ"


@pytest.mark.parametrize("wh", [(30, 18), (101, 7), (4, 4), (3, 5), (130, 3)])
def test_interpolate_ragged_sizes_exact(ctx, oracle, wh):
    """Widths that are not multiples of 4 (the per-thread pixel group) and non-power-of-two sizes,
    where uv*W - 0.5 is inexact and the bilinear weights drift off 0/1."""
    W, H = wh
    p, c = rand_frame(W, H), rand_frame(W, H)
    mv = np.zeros((H, W, 2), np.int8)
    mv[::2, ::3] = (1, 0)
    mv[1::4, 1::5] = (0, -1)
    got = run_interpolate(ctx, p, c, mv, 0.5)
    want = oracle.interpolate(p, c, mv.astype(np.float32), 0.5)
    assert (got == want).all()


def test_interpolate_table_cache_survives_many_sizes(oracle):
    """The per-size uv tables of the interpolate entry points are a bounded cache.  Round 3 evicted while a call already held
    the width's table (the height's lookup could free it): more sizes than the cache holds, each checked against the oracle,
    widths and heights arranged so that a call's width is the oldest entry when its height is new."""
    from linux_fg_amd import capi
    rng = np.random.default_rng(99)
    c = capi.Context(0)
    try:
        sizes = [(40 + 4 * i, 20 + i) for i in range(12)]           # 24 distinct axis lengths
        sizes += [(56, 70), (64, 71), (72, 72), (80, 73)]            # (56, 70): the width's table is the oldest cached entry, the height is new
        for (w, h) in sizes:
            prev = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
            curr = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
            mv = np.zeros((h, w, 2), np.int8)
            mv[h // 2:] = rng.integers(-2, 3, size=(h - h // 2, w, 2))
            got = run_interpolate(c, prev, curr, mv, 0.5)
            assert (got == oracle.interpolate(prev, curr, mv.astype(np.float32), 0.5)).all(), (w, h)
    finally:
        c.close()


def test_interpolate_kats(ctx):
    W, H = 64, 24
    p, c = rand_frame(W, H), rand_frame(W, H)
    zero = np.zeros((H, W, 2), np.int8)
    assert (run_interpolate(ctx, p, p, zero, 0.5) == p).all()                    # I-KAT3
    assert (run_interpolate(ctx, p, c, np.full((H, W, 2), -16, np.int8), 0.5) == 0).all()   # I-KAT2
    out = run_interpolate(ctx, p, c, zero, 0.25).astype(np.float64)              # I-KAT1
    want = p.astype(np.float64) * 0.75 + c.astype(np.float64) * 0.25
    assert (np.abs(out - want) <= 0.5 + 1e-3).all()


def test_interpolate_all_byte_values_round_trip(ctx):
    """prev == curr holding every byte value, MV = 0: out == in only if UNORM8 -> float -> UNORM8 is
    exact on the device for all 256 values (the two-op 1/255 form in csrc/lfg_device.hpp)."""
    vals = np.arange(256, dtype=np.uint8)
    f = np.stack([vals, vals[::-1], np.roll(vals, 7), np.roll(vals, 101)], -1)[None].repeat(8, 0)
    f = np.ascontiguousarray(f)
    zero = np.zeros((8, 256, 2), np.int8)
    for t in (0.25, 0.5, 0.75):
        assert (run_interpolate(ctx, f, f, zero, t) == f).all()


def test_interpolate_4k_exact_rois(ctx, oracle):
    W, H = 3840, 2160
    prev = synth.make_prev(W, H, seed=5)
    curr = synth.translate(prev, (3, -2), 5)
    mv = np.zeros((H, W, 2), np.int8)
    mv[:, W // 2:] = (-3, 2)
    got = run_interpolate(ctx, prev, curr, mv, 0.5)
    for roi in [(0, 0, 128, 16), (W // 2 - 32, 1000, W // 2 + 32, 1012), (W - 64, H - 8, W, H)]:
        want = oracle.interpolate(prev, curr, mv.astype(np.float32), 0.5, roi=roi)
        x0, y0, x1, y1 = roi
        assert (got[y0:y1, x0:x1] == want[y0:y1, x0:x1]).all(), roi
    assert (run_interpolate(ctx, prev, prev, np.zeros((H, W, 2), np.int8), 0.5) == prev).all()


@pytest.mark.parametrize("wh", [(3840, 2160), (1920, 1080), (7680, 4320), (1000, 333)])
def test_interpolate_every_pixel_where_it_samples(ctx, oracle, wh):
    """Content on which the stage really samples both frames (round 3 timed it on a pan, where the literal semantics reject
    every sample): zero vectors nearly everywhere -- the quad fast path, its row blends in the columns whose uv misses a texel
    centre in fp32 (102 of 3840, in runs) and the generic path in the rows that miss (85 of 2160) -- with islands of small
    non-zero vectors, at every benchmark size and a ragged one, EVERY pixel against the oracle, one factor and three."""
    W, H = wh
    rng = np.random.default_rng(W * 7 + H)
    prev = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    curr = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    mv = np.zeros((H, W, 2), np.int8)
    for _ in range(12):
        x0, y0 = int(rng.integers(0, W - 40)), int(rng.integers(0, H - 40))
        mv[y0:y0 + int(rng.integers(1, 40)), x0:x0 + int(rng.integers(1, 40))] = rng.integers(-1, 2, size=2)
    mvf = mv.astype(np.float32)
    for t in (0.5, 0.3):
        got = run_interpolate(ctx, prev, curr, mv, t)
        want = oracle.interpolate(prev, curr, mvf, t)
        assert (got == want).all(), f"t = {t}: {(got != want).any(-1).sum()} pixels differ, first at {np.argwhere((got != want).any(-1))[:4].tolist()}"
    factors = [0.25, 0.5, 0.75]
    for f, got in zip(factors, _run_interpolate_multi(ctx, prev, curr, mv, factors)):
        want = oracle.interpolate(prev, curr, mvf, f)
        assert (got == want).all(), f"one pass, t = {f}: {(got != want).any(-1).sum()} pixels differ"


def test_interpolate_fast_path_with_rows_that_are_only_4_byte_aligned(ctx, oracle):
    """The quad fast path reads 16 bytes per row and frame at row * pitch + 16 * quad: with a pitch of W * 4 + 4 those addresses
    are 4-byte aligned and no more (csrc/interpolate.hip, fast_frames: that is all it asks of prev and curr).  Static content at 4K --
    every pixel samples, the rows whose uv misses a texel centre (85 of 2160) take the generic path -- EVERY pixel against the
    oracle; the vectors and the output stay tightly packed, so the fast path is the one that runs."""
    from linux_fg_amd import capi
    W, H = 3840, 2160
    rng = np.random.default_rng(4242)
    prev = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    curr = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    mv = np.zeros((H, W, 2), np.int8)
    mv[1000:1030, 2000:2040] = (1, -1)
    bp, vp = _pitched(ctx, prev, 1)
    bc, vc = _pitched(ctx, curr, 1)
    assert vp.pitch == W * 4 + 4 and vp.pitch % 8 != 0
    m, out = ctx.frame_from(mv, capi.FORMAT_MV_S8X2), ctx.create_frame(W, H)
    for t in (0.5, 0.3):
        ctx.interpolate(vp, vc, m, out, t)
        got, want = ctx.download(out), oracle.interpolate(prev, curr, mv.astype(np.float32), t)
        assert (got == want).all(), f"t = {t}: {(got != want).any(-1).sum()} pixels differ, first at {np.argwhere((got != want).any(-1))[:4].tolist()}"
    for f in (bp, bc, m, out):
        ctx.destroy_frame(f)


# ------------------------------------------------------------------------------ whole path

def _run_interpolate_multi(ctx, prev, curr, mv_i8, factors):
    from linux_fg_amd import capi
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    m = ctx.frame_from(mv_i8, capi.FORMAT_MV_S8X2)
    outs = [ctx.create_frame(prev.shape[1], prev.shape[0]) for _ in factors]
    ctx.interpolate_multi(p, c, m, outs, factors)
    got = [ctx.download(o) for o in outs]
    for f in [p, c, m] + outs:
        ctx.destroy_frame(f)
    return got


@pytest.mark.parametrize("wh,factors", [((128, 40), [0.25, 0.5, 0.75]), ((101, 7), [0.5, 0.3]), ((90, 52), [0.25, 0.5, 0.75, 0.9]),
                                        ((66, 31), [0.1, 0.2, 0.4, 0.6, 0.8]), ((64, 16), [0.5])])
def test_interpolate_multi_equals_the_oracle_per_factor(ctx, oracle, wh, factors):
    """lfg_interpolate_multi (one pass, N frames: BASELINE config 5's t = 1/4, 1/2, 3/4) against the oracle's
    interpolate for every factor, bit for bit: motion vectors mixed from zero (both sources in range for every t),
    small ones (in range for some t only) and large ones (out of range), ragged widths, 1 to 5 factors (5 = 4 + 1)."""
    w, h = wh
    prev, curr = rand_frame(w, h), rand_frame(w, h)
    mv = RNG.integers(-16, 17, size=(h, w, 2)).astype(np.int8)
    mv[: h // 2, : w // 2] = 0
    mv[h // 2:, : w // 3] = RNG.integers(-1, 2, size=(h - h // 2, w // 3, 2)).astype(np.int8)
    got = _run_interpolate_multi(ctx, prev, curr, mv, factors)
    for g, t in zip(got, factors):
        assert (g == oracle.interpolate(prev, curr, mv.astype(np.float32), t)).all(), f"t = {t}"


def test_interpolate_multi_intended_semantics_and_validation(intended, oracle):
    from linux_fg_amd import capi
    w, h = 90, 52
    prev, curr = rand_frame(w, h), rand_frame(w, h)
    mv = RNG.integers(-16, 17, size=(h, w, 2)).astype(np.int8)
    mv[:10] = 0
    factors = [0.25, 0.5, 0.75]
    got = _run_interpolate_multi(intended, prev, curr, mv, factors)
    for g, t in zip(got, factors):
        one = run_interpolate(intended, prev, curr, mv, t)            # the single-factor kernel, same semantics
        assert (g == one).all()
        want = oracle.interpolate(prev, curr, mv.astype(np.float32), t, semantics=oracle.INTENDED)
        assert np.abs(g.astype(np.int16) - want.astype(np.int16)).max() <= 1
    a, b = intended.create_frame(w, h), intended.create_frame(w, h)
    m = intended.create_frame(w, h, capi.FORMAT_MV_S8X2)
    small = intended.create_frame(w // 2, h)
    out = intended.create_frame(w, h)
    with pytest.raises(capi.LfgError, match="alias each other"):
        intended.interpolate_multi(a, b, m, [out, out], [0.25, 0.5])
    with pytest.raises(capi.LfgError, match="bad output frame"):
        intended.interpolate_multi(a, b, m, [small], [0.5])
    with pytest.raises(capi.LfgError, match="aliases an input"):
        intended.interpolate_multi(a, b, m, [a], [0.5])
    with pytest.raises(capi.LfgError, match="count must be"):
        intended.interpolate_multi(a, b, m, [], [])
    for f in (a, b, m, small, out):
        intended.destroy_frame(f)


def test_interpolate_frames_multi_entry_point(ctx, oracle):
    """lfg_interpolate_frames_multi: motion once, then all factors -- against oracle.motion + oracle.interpolate."""
    w, h = 128, 72
    prev, curr = synth.make_pair(w, h, stream=2, shift=(0, 0))        # static pair: vectors are (-16,-16) or ties...
    curr = curr.copy(); curr[20:40, 30:90] = synth.noise_bytes(w, h, 5)[20:40, 30:90]
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    factors = [0.25, 0.5, 0.75]
    outs = [ctx.create_frame(w, h) for _ in factors]
    ctx.interpolate_frames_multi(p, c, outs, factors)
    mv = oracle.motion(prev, curr)
    for o, t in zip(outs, factors):
        assert (ctx.download(o) == oracle.interpolate(prev, curr, mv, t)).all()
    for f in [p, c] + outs:
        ctx.destroy_frame(f)


@pytest.mark.parametrize("wh,out_wh", [((96, 54), (192, 108)), ((250, 70), (500, 140)), ((64, 36), (100, 50)), ((40, 30), (40, 30))])
def test_interpolate_scale_equals_the_two_stages(ctx, oracle, wh, out_wh):
    """lfg_interpolate_scale (the input-resolution data flow): where out is exactly 2x the
    inputs ONE kernel interpolates every input row on the fly inside the 2x scale kernel -- and must give, byte for
    byte, what lfg_interpolate into a frame followed by lfg_scale of that frame gives (and within +-1 LSB what the
    oracle's two functions give); other size ratios take the two stages through a context-owned frame."""
    from linux_fg_amd import capi
    w, h = wh
    ow, oh = out_wh
    prev, curr = rand_frame(w, h), rand_frame(w, h)
    mv = RNG.integers(-16, 17, size=(h, w, 2)).astype(np.int8)
    mv[: h // 2] = 0                                              # both sources sampled in range: the bilinear path at texel centres
    mv[h // 2:, : w // 2] = RNG.integers(-1, 2, size=(h - h // 2, w // 2, 2)).astype(np.int8)
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    m = ctx.frame_from(mv, capi.FORMAT_MV_S8X2)
    mid, staged, fused = ctx.create_frame(w, h), ctx.create_frame(ow, oh), ctx.create_frame(ow, oh)
    for sem in (capi.SEMANTICS_REFERENCE, capi.SEMANTICS_INTENDED):
        ctx.set_semantics(sem)
        try:
            for t in (0.5, 0.25):
                ctx.interpolate(p, c, m, mid, t)
                ctx.scale(mid, staged)
                a = ctx.download(staged)
                for fuse in (True, False):                      # the fused kernel, and the default (two stages inside)
                    ctx.set_fused_interpolate_scale(fuse)
                    ctx.interpolate_scale(p, c, m, fused, t)
                    b = ctx.download(fused)
                    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ (semantics {sem}, t {t}, fused {fuse})"
                if sem == capi.SEMANTICS_REFERENCE:
                    want = oracle.scale(oracle.interpolate(prev, curr, mv.astype(np.float32), t), ow, oh)
                    assert_within_1lsb(b, want)
        finally:
            ctx.set_semantics(capi.SEMANTICS_REFERENCE)
            ctx.set_fused_interpolate_scale(False)
    other = ctx.create_frame(w + 2, h)
    with pytest.raises(capi.LfgError, match="differ in size"):
        ctx.interpolate_scale(p, other, m, fused, 0.5)
    ctx.destroy_frame(other)
    with pytest.raises(capi.LfgError, match="aliases an input"):
        ctx.interpolate_scale(p, c, m, c, 0.5)
    for f in (p, c, m, mid, staged, fused):
        ctx.destroy_frame(f)


def test_interpolate_scale_1080p_to_4k(ctx):
    """The benchmark's input-resolution variant at full size: fused kernel == the two stages, on the device's own motion vectors."""
    from linux_fg_amd import capi
    prev, curr = synth.make_pair(1920, 1080, stream=0, shift=(3, -2))
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    m = ctx.create_frame(1920, 1080, capi.FORMAT_MV_S8X2)
    ctx.motion(p, c, m)
    mid, staged, fused = ctx.create_frame(1920, 1080), ctx.create_frame(3840, 2160), ctx.create_frame(3840, 2160)
    ctx.interpolate(p, c, m, mid, 0.5)
    ctx.scale(mid, staged)
    ctx.set_fused_interpolate_scale(True)
    try:
        ctx.interpolate_scale(p, c, m, fused, 0.5)
    finally:
        ctx.set_fused_interpolate_scale(False)
    assert (ctx.download(staged) == ctx.download(fused)).all()
    for f in (p, c, m, mid, staged, fused):
        ctx.destroy_frame(f)


def test_interpolate_frames_entry_point(ctx, oracle):
    """FrameManager::InterpolateFrames equivalent: motion(8,16) then interpolate, MV temp inside."""
    W, H = 128, 64
    prev = synth.make_prev(W, H, seed=31)
    curr = synth.translate(prev, (0, 0), 31)
    curr[20:30, 40:60] = rand_frame(20, 10)
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    o = ctx.create_frame(W, H)
    ctx.interpolate_frames(p, c, o, 0.5)
    got = ctx.download(o)
    mv = oracle.motion(prev, curr)
    want = oracle.interpolate(prev, curr, mv, 0.5)
    assert (got == want).all()
    for f in (p, c, o):
        ctx.destroy_frame(f)


def test_interpolate_frames_in_the_north_star_order(ctx, oracle):
    """lfg_set_fused_motion_interpolate: the motion kernels write the generated frame from each vector
    as they decide it, the interpolate dispatch and the vector temporary are gone.  The frame has to be, byte for byte,
    what the two stages produce -- on content that takes every route to a vector: segments settled in the prefilter (a
    pan, a still), pooled rim segments, pixels the resolve kernel decides (noise, patches), tiles through the literal
    kernel (a fade), the literal kernel alone, both tie orders, ragged and pitched frames -- and what the oracle says."""
    from linux_fg_amd import capi
    cases = [(128, 64, 5), (1000, 350, 6), (777, 301, 7), (1920, 1080, 8)]
    for (w, h, seed) in cases:
        prev, curr = _mixed_pair(w, h, 5100 + seed)
        p, c = ctx.frame_from(prev), ctx.frame_from(curr)
        a, b = ctx.create_frame(w, h), ctx.create_frame(w, h)
        for mode in (capi.MOTION_PREFILTERED, capi.MOTION_EXACT_ONLY):
            for semantics in (capi.SEMANTICS_REFERENCE, capi.SEMANTICS_INTENDED):
                if (w, mode) == (1920, capi.MOTION_EXACT_ONLY) and semantics == capi.SEMANTICS_INTENDED:
                    continue                                     # (17 ms a call: one tie order is enough at this size)
                for t in (0.5, 0.25):
                    ctx.set_motion_mode(mode); ctx.set_semantics(semantics)
                    ctx.set_fused_motion_interpolate(False)
                    ctx.interpolate_frames(p, c, a, t)
                    ctx.set_fused_motion_interpolate(True)
                    ctx.upload(b, np.full((h, w, 4), 0xA5, np.uint8))       # (every pixel has to be written)
                    ctx.interpolate_frames(p, c, b, t)
                    ctx.set_fused_motion_interpolate(False)
                    assert (ctx.download(a) == ctx.download(b)).all(), (w, h, mode, semantics, t)
        ctx.set_motion_mode(capi.MOTION_PREFILTERED); ctx.set_semantics(capi.SEMANTICS_REFERENCE)
        if w <= 128:                                             # ... and the oracle, where it finishes in seconds
            ctx.set_fused_motion_interpolate(True)
            ctx.interpolate_frames(p, c, b, 0.5)
            ctx.set_fused_motion_interpolate(False)
            assert (ctx.download(b) == oracle.interpolate(prev, curr, oracle.motion(prev, curr), 0.5)).all()
        for f in (p, c, a, b):
            ctx.destroy_frame(f)
    # a still and a pan at 4K-class size: the settled-in-place route for nearly every pixel
    w, h = 2240, 1280
    prev = synth.make_prev(w, h, seed=5200)
    for curr in (prev.copy(), synth.translate(prev, (6, -4), 5201)):
        p, c = ctx.frame_from(prev), ctx.frame_from(curr)
        a, b = ctx.create_frame(w, h), ctx.create_frame(w, h)
        ctx.interpolate_frames(p, c, a, 0.5)
        ctx.set_fused_motion_interpolate(True)
        ctx.interpolate_frames(p, c, b, 0.5)
        ctx.set_fused_motion_interpolate(False)
        assert (ctx.download(a) == ctx.download(b)).all()
        for f in (p, c, a, b):
            ctx.destroy_frame(f)


def test_north_star_order_with_frames_in_flight():
    """The fused order on a context with three lanes (its own work-unit plan, the lanes' workspaces, the launches sized by each
    lane's previous call): the generated frame is, byte for byte, the two stages' on every lane -- a pan, a mixture of content
    kinds, and the pan again, so that a lane's verdict from one content meets the next."""
    from linux_fg_amd import capi
    w, h = 1920, 1080
    base = synth.make_prev(w, h, seed=5300)
    pairs = [(base, synth.translate(base, (4, -3), 5301)), _mixed_pair(w, h, 5302), (base, synth.translate(base, (4, -3), 5301))]
    c = capi.Context(0)
    try:
        c.lanes(3)
        for k, (prev, curr) in enumerate(pairs * 2):
            c.lane_select(k % 3)
            p, q = c.frame_from(prev), c.frame_from(curr)
            a, b = c.create_frame(w, h), c.create_frame(w, h)
            c.set_fused_motion_interpolate(False)
            c.interpolate_frames(p, q, a, 0.5)
            c.set_fused_motion_interpolate(True)
            c.upload(b, np.full((h, w, 4), 0xA5, np.uint8))
            c.interpolate_frames(p, q, b, 0.5)
            c.set_fused_motion_interpolate(False)
            assert (c.download(a) == c.download(b)).all(), (k, k % 3)
            for f in (p, q, a, b):
                c.destroy_frame(f)
        c.lane_select(0)
    finally:
        c.close()


def test_three_stage_path_small(ctx, oracle):
    """north_star order at small size: scale(prev), scale(curr) -> motion -> interpolate, each stage
    fed with the DEVICE result of the previous one; oracle chained the same way from the device's
    scaled frames (scale itself is checked to +-1 LSB above)."""
    from linux_fg_amd import capi
    w, h = 64, 36
    prev, curr = synth.make_pair(w, h, stream=2, shift=(1, 1))
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    P, C, O = ctx.create_frame(2 * w, 2 * h), ctx.create_frame(2 * w, 2 * h), ctx.create_frame(2 * w, 2 * h)
    M = ctx.create_frame(2 * w, 2 * h, capi.FORMAT_MV_S8X2)
    ctx.scale(p, P); ctx.scale(c, C)
    ctx.motion(P, C, M)
    ctx.interpolate(P, C, M, O, 0.5)
    Pn, Cn, Mn, On = ctx.download(P), ctx.download(C), ctx.download(M), ctx.download(O)
    assert_within_1lsb(Pn, oracle.scale(prev, 2 * w, 2 * h))
    assert_within_1lsb(Cn, oracle.scale(curr, 2 * w, 2 * h))
    mv = oracle.motion(Pn, Cn)
    assert (Mn == mv.astype(np.int8)).all()
    assert (On == oracle.interpolate(Pn, Cn, mv, 0.5)).all()
    for f in (p, c, P, C, O, M):
        ctx.destroy_frame(f)


# ------------------------------------------------------------------------------ golden fixtures

def test_golden_fixtures_on_device(ctx):
    """tests/golden/golden_small.npz (oracle outputs on seeded frames, committed): the device path
    reproduces them without the oracle being rebuilt or run."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_small.npz"))
    assert_within_1lsb(run_scale(ctx, g["prev_in"], 128, 72), g["prev_up"])
    assert_within_1lsb(run_scale(ctx, g["curr_in"], 128, 72), g["curr_up"])
    assert_within_1lsb(run_scale(ctx, g["curr_in"], 53, 41), g["curr_53x41"])
    assert (run_motion(ctx, g["prev_up"], g["curr_up"]) == g["mv"]).all()
    assert (run_motion(ctx, g["prev_in"], g["curr_in"], 4, 3.0) == g["mv_b4_r3"]).all()
    for t in (25, 50, 75):
        assert (run_interpolate(ctx, g["prev_up"], g["curr_up"], g["mv"], t / 100.0) == g[f"interp_{t}"]).all()


# ------------------------------------------------------------------------------ pitched frames

def _pitched(ctx, host, pad_px, fmt=None):
    """Upload `host` into the left part of a wider device allocation and describe it with a row pitch
    larger than width * bpp (lfg_frame_wrap), as a caller handing over a sub-rectangle would."""
    from linux_fg_amd import capi
    fmt = capi.FORMAT_RGBA8 if fmt is None else fmt
    h, w, ch = host.shape
    wide = np.zeros((h, w + pad_px, ch), host.dtype)
    wide[:, :w] = host
    wide[:, w:] = 0x5A if host.dtype == np.uint8 else 3          # poison the padding
    big = ctx.frame_from(wide, fmt)
    view = capi.Context.wrap(big.data, w, h, fmt, pitch=(w + pad_px) * ch)
    return big, view


def _read_pitched(ctx, big, w):
    return ctx.download(big)[:, :w]


@pytest.mark.parametrize("pad", [4, 12])
def test_all_stages_with_row_pitch_larger_than_width(ctx, oracle, pad):
    from linux_fg_amd import capi
    w, h = 64, 36
    prev, curr = synth.make_pair(w, h, stream=4, shift=(-2, 1))
    bp, vp = _pitched(ctx, prev, pad)
    bc, vc = _pitched(ctx, curr, pad)
    W, H = 2 * w, 2 * h
    bP, vP = _pitched(ctx, np.zeros((H, W, 4), np.uint8), pad)
    bC, vC = _pitched(ctx, np.zeros((H, W, 4), np.uint8), pad)
    bO, vO = _pitched(ctx, np.zeros((H, W, 4), np.uint8), pad)
    bM, vM = _pitched(ctx, np.zeros((H, W, 2), np.int8), pad, capi.FORMAT_MV_S8X2)
    ctx.scale(vp, vP); ctx.scale(vc, vC)
    ctx.motion(vP, vC, vM)
    ctx.interpolate(vP, vC, vM, vO, 0.5)
    Pn, Cn, Mn, On = (_read_pitched(ctx, b, W) for b in (bP, bC, bM, bO))
    assert_within_1lsb(Pn, oracle.scale(prev, W, H))
    assert_within_1lsb(Cn, oracle.scale(curr, W, H))
    mv = oracle.motion(Pn, Cn)
    assert (Mn == mv.astype(np.int8)).all()
    assert (On == oracle.interpolate(Pn, Cn, mv, 0.5)).all()
    for b in (bP, bC, bO):                                   # nothing was written into the padding
        assert (ctx.download(b)[:, W:] == 0x5A).all()
    assert (ctx.download(bM)[:, W:] == 3).all()
    for b in (bp, bc, bP, bC, bO, bM):
        ctx.destroy_frame(b)


def test_frame_copy_moves_the_bytes(ctx):
    """lfg_frame_copy (FrameManager::CopyFrameData, src/frame_manager.cpp:83-145): tight -> tight, pitched -> tight,
    tight -> pitched (padding untouched) and motion-vector frames."""
    from linux_fg_amd import capi
    w, h = 70, 33
    a = rand_frame(w, h)
    src, dst = ctx.frame_from(a), ctx.create_frame(w, h)
    ctx.copy(src, dst)
    assert (ctx.download(dst) == a).all()
    big, view = _pitched(ctx, a, 12)                              # pitched source
    dst2 = ctx.frame_from(np.zeros_like(a))
    ctx.copy(view, dst2)
    assert (ctx.download(dst2) == a).all()
    bigd, viewd = _pitched(ctx, np.zeros_like(a), 4)              # pitched destination
    ctx.copy(src, viewd)
    got = ctx.download(bigd)
    assert (got[:, :w] == a).all() and (got[:, w:] == 0x5A).all()
    m = RNG.integers(-16, 17, size=(h, w, 2)).astype(np.int8)
    ms, md = ctx.frame_from(m, capi.FORMAT_MV_S8X2), ctx.create_frame(w, h, capi.FORMAT_MV_S8X2)
    ctx.copy(ms, md)
    assert (ctx.download(md) == m).all()
    with pytest.raises(capi.LfgError, match="dimensions do not match"):
        ctx.copy(ms, dst)                                         # same size, different format
    for f in (src, dst, big, dst2, bigd, ms, md):
        ctx.destroy_frame(f)


def test_mv_export_is_the_reference_rgba32f_image(ctx, oracle):
    """lfg_mv_export_rgba32f reproduces what shaders/motion.comp:56 stores -- vec4(best, 0, 1) in an rgba32f image --
    bit for bit from the oracle's float vectors, for a tight and a pitched motion-vector frame."""
    from linux_fg_amd import capi
    w, h = 96, 50
    prev, curr = synth.make_pair(w, h, stream=3, shift=(4, -7))
    want = oracle.motion(prev, curr)                              # (H, W, 2) float32
    p, c = ctx.frame_from(prev), ctx.frame_from(curr)
    mv = ctx.create_frame(w, h, capi.FORMAT_MV_S8X2)
    ctx.motion(p, c, mv)
    img = ctx.mv_export_rgba32f(mv)
    assert img.dtype == np.float32 and img.shape == (h, w, 4)
    assert (img[..., :2].view(np.uint32) == want.view(np.uint32)).all()
    assert (img[..., 2] == 0.0).all() and (img[..., 3] == 1.0).all()
    big, view = _pitched(ctx, ctx.download(mv), 6, capi.FORMAT_MV_S8X2)
    assert (ctx.mv_export_rgba32f(view) == img).all()
    for f in (p, c, mv, big):
        ctx.destroy_frame(f)


def test_motion_workspace_size_and_list_depths_on_pure_noise(ctx):
    """lfg_motion_workspace_size: what the prefiltered path keeps per lane (at 4K 0.90 GB with frames in flight, 1.00 GB for a
    context that runs one frame at a time and searches the top and bottom strips in eight parts; list depth 10 -- round 2:
    2.2 / 2.4 GB at depths 32 / 24 / 24, before a pixel's count restarted whenever a candidate undercut its threshold by more
    than the bracket's width), refusals; and the depth is enough where lists used to be longest -- a pair of independent noise
    frames, every pixel without a match, 7.5 records per pixel over the search of which a list holds the last one or two: no
    tile overflows into the literal kernel, and the vectors are the literal kernel's."""
    from linux_fg_amd import capi
    n4k = ctx.motion_workspace_size(3840, 2160)
    assert 0.3e9 < n4k < 1.0e9
    other = capi.Context(0)
    try:
        other.lanes(2)
        # (frames in flight: two plans side by side since round 4 -- rim segments in four parts, and the plan of the calls that go
        #  through the lean kernel, whose top and bottom strips are searched in eight; the auxiliary lists serve either)
        assert other.motion_workspace_size(3840, 2160) <= n4k + (1 << 20)      # (+ the second plan's tables and the lean kernel's tile list)
    finally:
        other.close()
    assert ctx.motion_workspace_size(1920, 1080) < n4k
    for bad in ((0, 16), (16, 0), (40000, 16)):
        with pytest.raises(capi.LfgError):
            ctx.motion_workspace_size(*bad)
    w, h = 1920, 1080
    prev = synth.noise_bytes(w, h, 4242)
    curr = synth.noise_bytes(w, h, 4243)
    a, st = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all()
    assert st[1] == 0, f"{st[1]} of {st[0]} tiles overflowed their lists"


def test_rim_segments_in_eight_parts_agree_with_the_literal_kernel(monkeypatch):
    """LFG_MOTION_RIM_SPLIT=8 (a knob for pan-heavy, latency-bound use: the strip a pan exposes is searched by two workgroups
    per segment instead of one): the plan, the private lists and the resolve kernel's eight-list merge against the literal
    kernel on a frame with rim, interior, occlusions and noise; and the workspace the query reports grows with it."""
    from linux_fg_amd import capi
    base = capi.Context(0)
    try:
        n4 = base.motion_workspace_size(1940, 1090)
    finally:
        base.close()
    monkeypatch.setenv("LFG_MOTION_RIM_SPLIT", "8")
    c8 = capi.Context(0)
    try:
        assert c8.motion_workspace_size(1940, 1090) > n4       # (a frame with more tiles than half the resident workgroups: rim tiles go by segment)
        for seed in (31, 32):
            prev, curr = _mixed_pair(1940, 1090, 9700 + seed)
            a, _ = run_motion_mode(c8, prev, curr, capi.MOTION_PREFILTERED)
            b, _ = run_motion_mode(c8, prev, curr, capi.MOTION_EXACT_ONLY)
            assert (a == b).all()
    finally:
        c8.close()


def test_staging_buffers_round_trip(ctx):
    """lfg_staging_create / lfg_staging_destroy (FrameManager::CreateStagingBuffer, src/frame_manager.cpp:199-214):
    pinned host memory, asynchronous upload from it and read-back into it."""
    w, h = 200, 64
    a = rand_frame(w, h)
    up, down = ctx.staging_create(a.nbytes), ctx.staging_create(a.nbytes)
    up[:] = a.reshape(-1)
    down[:] = 0
    f, g = ctx.create_frame(w, h), ctx.create_frame(w, h)
    ctx.upload_async(f, up)
    ctx.copy(f, g)
    ctx.download_async(g, down)
    ctx.sync()
    assert (down.reshape(h, w, 4) == a).all()
    ctx.staging_destroy(up); ctx.staging_destroy(down)
    ctx.staging_destroy(ctx.staging_create(1))                    # smallest size; destroy is the only owner
    with pytest.raises(capi_error(), match="bad argument"):
        ctx.staging_create(0)
    for fr in (f, g):
        ctx.destroy_frame(fr)


def capi_error():
    from linux_fg_amd import capi
    return capi.LfgError


def test_motion_frames_of_2_gib_and_more_take_the_generic_kernel(ctx, oracle):
    """The 8/16 kernels address with 32-bit byte offsets; a frame whose height x pitch reaches 2 GiB must not reach
    them (it would alias rows silently).  A 64 x 40 view with rows 57.6 MB apart spans 2.3 GB: lfg_motion routes it
    to the size_t-indexed generic kernel and the vectors still equal the oracle's."""
    from linux_fg_amd import capi
    w, h, pitch = 64, 40, 57_600_000
    prev, curr = synth.make_pair(w, h, stream=6, shift=(-3, 2))
    bigs, views = [], []
    for host in (prev, curr):
        big = ctx.create_frame(24000, 24000)                      # 2.304e9 bytes >= 39 * pitch + 256
        v = capi.Context.wrap(big.data, w, h, capi.FORMAT_RGBA8, pitch=pitch)
        assert v.height * v.pitch >= 2 ** 31
        ctx.upload(v, host)
        bigs.append(big); views.append(v)
    mv = ctx.create_frame(w, h, capi.FORMAT_MV_S8X2)
    ctx.motion(views[0], views[1], mv)
    assert (ctx.download(mv) == as_int(oracle.motion(prev, curr))).all()
    for f in bigs + [mv]:
        ctx.destroy_frame(f)


def test_argument_validation_on_device(ctx):
    from linux_fg_amd import capi
    a = ctx.create_frame(32, 16)
    b = ctx.create_frame(64, 32)
    m = ctx.create_frame(32, 16, capi.FORMAT_MV_S8X2)
    with pytest.raises(capi.LfgError, match="differ in size"):
        ctx.motion(a, b, m)
    with pytest.raises(capi.LfgError, match="searchRadius"):
        ctx.motion(a, a, m, 8, 2.5)
    with pytest.raises(capi.LfgError, match="blockSize"):
        ctx.motion(a, a, m, 0, 16.0)
    with pytest.raises(capi.LfgError, match="wrong format|bad frame"):
        ctx.interpolate(a, a, a, a, 0.5)
    with pytest.raises(capi.LfgError, match="dimensions do not match"):
        ctx.copy(a, b)
    with pytest.raises(capi.LfgError, match="smaller than expected"):
        ctx._check(ctx.lib.lfg_frame_upload(ctx.h, a, np.zeros(8, np.uint8).ctypes.data, 8), "lfg_frame_upload")
    for f in (a, b, m):
        ctx.destroy_frame(f)


# ------------------------------------------------------------------ opt-in "intended" semantics

@pytest.fixture()
def intended(ctx):
    from linux_fg_amd import capi
    ctx.set_semantics(capi.SEMANTICS_INTENDED)
    yield ctx
    ctx.set_semantics(capi.SEMANTICS_REFERENCE)


def test_intended_motion_kats(intended, oracle):
    """Zero and flat frames report (0,0) instead of (-16,-16); a pure translation still reports -t."""
    from linux_fg_amd import capi
    z = np.zeros((40, 72, 4), np.uint8)
    f = np.empty((70, 130, 4), np.uint8); f[...] = (40, 90, 200, 255)
    for mode in (capi.MOTION_PREFILTERED, capi.MOTION_EXACT_ONLY):
        assert (run_motion_mode(intended, z, z, mode)[0] == 0).all()
        assert (run_motion_mode(intended, f, f, mode)[0] == 0).all()
    prev = synth.make_prev(160, 90, seed=9)
    curr = synth.translate(prev, (5, -3), seed=9)
    got, _ = run_motion_mode(intended, prev, curr, capi.MOTION_PREFILTERED)
    assert (got[24:-24, 24:-24, 0] == -5).all() and (got[24:-24, 24:-24, 1] == 3).all()


@pytest.mark.parametrize("name", ["noise", "periodic", "flat + noise", "static with flat areas"])
def test_intended_motion_matches_oracle(intended, oracle, name):
    """Both 8/16 paths against the oracle's intended mode on the tie-heavy adversarial frames; the two modes
    differ from the reference exactly where costs tie."""
    from linux_fg_amd import capi
    prev, curr = _adversarial_pairs()[name]
    want = as_int(oracle.motion(prev, curr, semantics=oracle.INTENDED))
    for mode in (capi.MOTION_PREFILTERED, capi.MOTION_EXACT_ONLY):
        got, _ = run_motion_mode(intended, prev, curr, mode)
        assert (got == want).all(), f"{(got != want).any(-1).sum()} pixels differ ({name}, mode {mode})"
    if name in ("flat + noise", "static with flat areas"):
        assert (want != as_int(oracle.motion(prev, curr))).any()


def test_intended_motion_large_frame_mechanisms(intended):
    """1080p frames, where tiles are whole work units: hand-over (noise patches), four-point sums (sensor noise),
    plateaus at the rim and flat areas with ties, all under the intended tie order -- prefiltered path == literal
    kernel."""
    from linux_fg_amd import capi
    W, H = 1920, 1080
    prev = synth.make_prev(W, H, seed=synth.BASE_SEED + 33)
    curr = synth.translate(prev, (6, -9), synth.BASE_SEED + 33)
    n = synth.noise_bytes(W, H, 99) % 5
    curr = np.clip(curr.astype(np.int16) + n.astype(np.int16) - 2, 0, 255).astype(np.uint8)
    fresh = synth.noise_bytes(W, H, 4711)
    for (x0, y0, w, h) in [(300, 200, 120, 90), (1000, 500, 64, 200), (1500, 900, 250, 40)]:
        curr[y0:y0 + h, x0:x0 + w] = fresh[y0:y0 + h, x0:x0 + w]
    prev[600:700, 100:400] = 37                            # a flat area, static: ties at cost 0
    curr[600:700, 100:400] = 37
    a, stats = run_motion_mode(intended, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(intended, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all(), f"{(a != b).any(-1).sum()} pixels differ"
    assert (a[640:660, 200:300] == 0).all()                # the flat area reports the shortest vector


def test_intended_motion_generic_parameters(intended, oracle):
    prev = synth.make_prev(50, 34, seed=21)
    prev[10:30, 5:40] = (7, 7, 7, 255)                                   # a flat patch: ties
    curr = prev.copy()
    for bs, R in [(4, 3.0), (8, 2.0), (5, 4.0)]:
        got = run_motion(intended, prev, curr, bs, R)
        assert (got == as_int(oracle.motion(prev, curr, bs, R, semantics=oracle.INTENDED))).all()


@pytest.mark.parametrize("t", [0.25, 0.5, 0.75])
def test_intended_interpolate_matches_oracle(intended, oracle, t):
    """Motion vectors now displace by pixels: fractional sample positions, the bilinear path for real."""
    w, h = 90, 52
    prev, curr = rand_frame(w, h), rand_frame(w, h)
    mv = RNG.integers(-16, 17, size=(h, w, 2)).astype(np.int8)
    got = run_interpolate(intended, prev, curr, mv, t)
    want = oracle.interpolate(prev, curr, mv.astype(np.float32), t, semantics=oracle.INTENDED)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    assert d.max() <= 1, f"max diff {d.max()}"
    assert (d != 0).mean() < 1e-3                                         # in practice identical
    lit = oracle.interpolate(prev, curr, mv.astype(np.float32), t)
    assert (want != lit).any()


def test_intended_interpolate_full_hd_every_pixel(intended, oracle):
    """The same at 1920x1080 with a smooth field of vectors plus zero areas: under the intended semantics every pixel samples
    displaced, mostly fractional positions -- the generic path is the whole frame here."""
    W, H = 1920, 1080
    rng = np.random.default_rng(77)
    prev = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    curr = rng.integers(0, 256, size=(H, W, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:H, 0:W]
    mv = np.stack([np.rint(9 * np.sin(xx / 211.0 + yy / 97.0)), np.rint(7 * np.cos(xx / 131.0 - yy / 173.0))], -1).astype(np.int8)
    mv[300:500, 600:900] = 0
    for t in (0.5, 0.25):
        got = run_interpolate(intended, prev, curr, mv, t)
        want = oracle.interpolate(prev, curr, mv.astype(np.float32), t, semantics=oracle.INTENDED)
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert d.max() <= 1, f"max diff {d.max()}"
        assert (d != 0).mean() < 1e-3


def test_lanes_keep_frames_in_flight_apart():
    """Lanes (include/linuxfg_hip.h): a stream of frames run with two and three frames in flight -- frame k on lane
    k % n with its own curr / mv / out buffers, a frame's motion waiting for the previous frame's upscale on the other
    lane -- gives, frame for frame, what one lane gives.  Sizes with more than one prefilter tile so that the lanes'
    motion workspaces are really in use at the same time."""
    from linux_fg_amd import capi
    ctx = capi.Context(0)
    try:
        assert ctx.lane_count() == 1 and ctx.lane_current() == 0
        with pytest.raises(capi.LfgError, match="no such lane"):
            ctx.lane_select(1)
        for bad in (0, capi.MAX_LANES + 1):
            with pytest.raises(capi.LfgError, match="count must be"):
                ctx.lanes(bad)
        w, h, n_frames = 160, 96, 7
        seed = synth.BASE_SEED + 77
        inputs = [synth.make_prev(w, h, seed)]
        for k in range(1, n_frames):
            inputs.append(synth.translate(inputs[-1], (2 + (k % 3), -1 - (k % 2)), seed + k))
        f_in = [ctx.frame_from(a) for a in inputs]

        def run(lanes):
            ctx.lanes(lanes)
            assert ctx.lane_count() == lanes
            ups = [ctx.create_frame(2 * w, 2 * h) for _ in range(n_frames)]
            mvs = [ctx.create_frame(2 * w, 2 * h, capi.FORMAT_MV_S8X2) for _ in range(n_frames)]
            outs = [ctx.create_frame(2 * w, 2 * h) for _ in range(n_frames)]
            for k in range(n_frames):
                ctx.lane_select(k % lanes)
                assert ctx.lane_current() == k % lanes
                ctx.lane_wait((k - 1) % lanes)              # (before the first mark of that lane: nothing to wait for)
                ctx.scale(f_in[k], ups[k])
                ctx.lane_mark()
                if k:
                    ctx.motion(ups[k - 1], ups[k], mvs[k], 8, 16.0)
                    ctx.interpolate(ups[k - 1], ups[k], mvs[k], outs[k], 0.5)
            ctx.sync()                                      # every lane
            got = [(ctx.download(mvs[k]), ctx.download(outs[k])) for k in range(1, n_frames)]
            ctx.lane_select(0)
            for f in ups + mvs + outs:
                ctx.destroy_frame(f)
            return got

        one = run(1)
        assert len({m.tobytes() for m, _ in one}) > 1       # the frames differ from each other
        for lanes in (2, 3):
            for (m1, o1), (m2, o2) in zip(one, run(lanes)):
                assert (m1 == m2).all() and (o1 == o2).all()
        ctx.lanes(1)                                        # shrinking frees the other lanes' workspaces
        assert ctx.lane_count() == 1
    finally:
        ctx.close()


# ------------------------------------------------------------------ the strip kernel (csrc/motion_strip.hip)
# The bands a pan exposes along the frame's edges, decided completely by a kernel of their own.  Whatever it takes, the vectors
# are the literal kernel's -- and the ORACLE's on the strips themselves.

def _strip_pairs():
    W, H = 700, 420
    base = synth.make_prev(W, H, seed=synth.BASE_SEED + 808)
    for shift in [(6, -4), (-6, 4), (5, 0), (0, 7), (-16, 16), (16, -3), (-1, -1), (9, 12)]:
        yield f"pan {shift}", base, synth.translate(base, shift, synth.BASE_SEED + 808), shift
    # the exposed strip DARK (zeros) instead of fresh noise: every candidate whose block leaves prev ties exactly there (plateaus),
    # and a black strip against black out-of-image texels costs exactly nothing
    dark = synth.translate(base, (7, -5), synth.BASE_SEED + 809)
    dark[:, :7] = 0; dark[-5:, :] = 0
    yield "pan, exposed strip black", base, dark, (7, -5)
    # letterbox: black bars above and below in both frames, a horizontal pan between them
    lb_prev, lb_curr = base.copy(), synth.translate(base, (-8, 0), synth.BASE_SEED + 810)
    for f in (lb_prev, lb_curr):
        f[:60] = 0; f[-60:] = 0
    yield "letterbox", lb_prev, lb_curr, (-8, 0)
    # a flat exposed strip one level away from the flat prev border: exact ties at a non-zero cost (lists overflow: the literal kernel's)
    flat_prev, flat_curr = base.copy(), synth.translate(base, (4, 0), synth.BASE_SEED + 811)
    flat_prev[:, :40] = 100; flat_curr[:, :44] = 101
    yield "flat border one level apart", flat_prev, flat_curr, (4, 0)


@pytest.mark.parametrize("case", range(11))
@pytest.mark.parametrize("lanes", [1, 3])
def test_strip_kernel_agrees_with_the_literal_kernel_and_the_oracle(oracle, monkeypatch, case, lanes):
    from linux_fg_amd import capi
    monkeypatch.setenv("LFG_MOTION_STRIP", "1")           # (opt-in: read when the context is created)
    name, prev, curr, shift = list(_strip_pairs())[case]
    H, W = prev.shape[:2]
    c = capi.Context(0)
    try:
        if lanes > 1:
            c.lanes(lanes)
        for sem in (capi.SEMANTICS_REFERENCE, capi.SEMANTICS_INTENDED):
            if sem == capi.SEMANTICS_INTENDED and case not in (0, 4, 8):
                continue
            c.set_semantics(sem)
            a, _ = run_motion_mode(c, prev, curr, capi.MOTION_PREFILTERED)
            rows, cols = c.motion_strip_stats()
            # the edges the translation exposes (the left or the right one, the top or the bottom one) -- as the call's top hint says
            assert rows == (H if shift[0] != 0 else 0) and cols == (W if shift[1] != 0 else 0), (name, rows, cols)
            b, _ = run_motion_mode(c, prev, curr, capi.MOTION_EXACT_ONLY)
            assert (a == b).all(), f"{name}, {lanes} lanes, semantics {sem}: {(a != b).any(-1).sum()} pixels differ from the literal kernel"
            # the oracle on the strips themselves: left or right band across an item boundary (57 rows), top or bottom band across one (57 columns), a corner
            rois = []
            if shift[0] != 0:
                x0 = 0 if shift[0] > 0 else W - 24
                rois += [(x0, 50, x0 + 24, 64), (x0, H - 14, x0 + 24, H)]
            if shift[1] != 0:
                y0 = 0 if shift[1] > 0 else H - 12
                rois += [(50, y0, 64, y0 + 12), (W - 20, y0, W, y0 + 12)]
            for roi in rois:
                rx0, ry0, rx1, ry1 = roi
                want = as_int(oracle.motion(prev, curr, roi=roi, semantics=oracle.INTENDED if sem == capi.SEMANTICS_INTENDED else oracle.REFERENCE))[ry0:ry1, rx0:rx1]
                assert (a[ry0:ry1, rx0:rx1] == want).all(), f"{name}, semantics {sem}: the oracle says otherwise in {roi}"
        c.set_semantics(capi.SEMANTICS_REFERENCE)
    finally:
        c.close()


def test_strip_kernel_at_4k_and_8k_against_the_oracle_on_the_strips(oracle, monkeypatch):
    """BASELINE config 3's and config 5's sizes, the benchmark's pan (both frames upscaled on the device), frames in flight: the
    oracle ON the strips the pan exposes -- the left band across two items, the bottom band across two items, the corner where they
    meet, the rows just above the bottom band (the persistent kernel's: their blocks reach into the strip) -- and on a tile the lean
    kernel LEFT (a moving object's edge inside the frame)."""
    from linux_fg_amd import capi
    monkeypatch.setenv("LFG_MOTION_STRIP", "1")
    c = capi.Context(0)
    try:
        c.lanes(3)
        for (w, h), rois_of in (((1920, 1080), lambda W, H: [(0, 1130, 24, 1150), (0, 560, 20, 580), (1130, H - 12, 1150, H), (2270, H - 20, 2290, H - 8),
                                                            (0, H - 16, 40, H), (W - 24, 40, W, 52)]),
                                ((3840, 2160), lambda W, H: [(0, 2270, 24, 2286), (4550, H - 12, 4570, H), (0, H - 12, 32, H)])):
            pin = synth.make_prev(w, h, seed=synth.BASE_SEED)
            cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
            if w == 1920:                       # a patch that moves on its own: the lean kernel leaves the tiles its edge crosses
                cin[500:560, 800:880] = pin[497:557, 804:884]
            p, q = c.frame_from(pin), c.frame_from(cin)
            W, H = 2 * w, 2 * h
            P, C = c.create_frame(W, H), c.create_frame(W, H)
            M = c.create_frame(W, H, capi.FORMAT_MV_S8X2)
            c.scale(p, P); c.scale(q, C)
            Pn, Cn = c.download(P), c.download(C)
            for lane in range(3):               # (every lane's workspace; the second call of a lane goes by its first call's verdict: the lean kernel)
                for _ in range(2):
                    c.lane_select(lane)
                    c.motion(P, C, M)
                    c.lane_sync()
            Mn = as_int(c.download(M))
            rows, cols = c.motion_strip_stats()
            assert rows == H and cols == W, (rows, cols)
            used, tiles, left = c.motion_lean_stats()
            assert used, "the lean kernel did not run"
            rois = rois_of(W, H)
            if w == 1920:
                assert left > 0
                rois.append((1590, 990, 1620, 1006))                    # the moving patch's left edge (4K coordinates)
            for roi in rois:
                x0, y0, x1, y1 = roi
                want = as_int(oracle.motion(Pn, Cn, roi=roi))[y0:y1, x0:x1]
                assert (Mn[y0:y1, x0:x1] == want).all(), f"{W}x{H}: vectors differ from the oracle in {roi}: {(Mn[y0:y1, x0:x1] != want).any(-1).sum()} pixels"
            for f in (p, q, P, C, M):
                c.destroy_frame(f)
        c.lane_select(0)
    finally:
        c.close()


# ------------------------------------------------------------------ the lean kernel (csrc/motion_lean.hip)

@pytest.fixture()
def lean_ctx(monkeypatch):
    """A context with frames in flight whose every call goes through the lean kernel (the host's choice per call -- the order
    kernel's verdict on the lane's previous call -- overridden by LFG_LEAN_FORCE=1), and a second plan beside the first."""
    from linux_fg_amd import capi
    monkeypatch.setenv("LFG_LEAN_FORCE", "1")
    c = capi.Context(0)
    c.lanes(2)
    yield c
    c.close()


def _lean_cases():
    W, H = 1920, 1080
    base = synth.make_prev(W, H, seed=synth.BASE_SEED + 404)
    pan = synth.translate(base, (5, -3), synth.BASE_SEED + 404)
    yield "pan", base, pan
    yield "stills", base, base.copy()
    rough = pan.copy()                       # a pan whose match is only nearly exact in a band: small non-zero thresholds, SAD and distance tests
    rough[200:420] = np.clip(rough[200:420].astype(np.int16) + (synth.noise_bytes(W, 220, 99) % 3).astype(np.int16) - 1, 0, 255).astype(np.uint8)
    yield "pan with a band one level off", base, rough
    yield "everything the prefilter treats differently", *_full_hd_mixture()
    flat = np.full((H, W, 4), 90, np.uint8)  # exact ties everywhere: every pixel ambiguous or zero-cost, the kernel must leave what it cannot decide
    yield "flat", flat, flat.copy()
    per = np.tile(synth.make_prev(16, 16, seed=5), (H // 16 + 1, W // 16 + 1, 1))[:H, :W].copy()   # periodic: many exact zero-cost candidates
    yield "periodic", per, np.roll(per, (3, 5), (0, 1))


@pytest.mark.parametrize("case", range(6))
def test_lean_kernel_agrees_with_the_literal_kernel(lean_ctx, case):
    """Whole interior tiles go through csrc/motion_lean.hip first when frames are in flight: it settles what is easy and LEAVES
    the rest to the persistent kernel, so the vectors must be the literal kernel's whatever it decides -- on content it is made
    for (a pan, stills), on content it must give up on segment by segment (noise bands, occlusions, fades: the mixture), and on
    exact ties (flat and periodic frames: zero-cost candidates decided by rank, or left)."""
    from linux_fg_amd import capi
    name, prev, curr = list(_lean_cases())[case]
    for lane in (0, 1):                      # both lanes' workspaces and plans
        lean_ctx.lane_select(lane)
        a, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_PREFILTERED)
        used, listed, left = lean_ctx.motion_lean_stats()
        assert used and listed > 300, (used, listed)          # (the kernel did run: 35 x 17 tiles of which the rim's are not its)
        if name in ("pan", "stills"):
            assert left == 0, f"{name}: the lean kernel left work in {left} of {listed} tiles"
        if name == "everything the prefilter treats differently":
            assert 0 < left < listed, (left, listed)
        b, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_EXACT_ONLY)
        assert (a == b).all(), f"{name}, lane {lane}: {(a != b).any(-1).sum()} pixels differ"
    lean_ctx.lane_select(0)


@pytest.mark.parametrize("case", [0, 1, 3, 4, 5])
def test_lean_kernel_under_the_intended_tie_order(lean_ctx, oracle, case):
    """Round 5: the lean kernel also serves lfg_set_semantics(INTENDED), where a candidate's rank is its place in the shortest-vector-
    first order and not its scan index (its ranks go through rank2scan): whole frames against the literal kernel under the same
    semantics -- ties (stills, flat and periodic frames) are exactly where the two orders disagree -- and a region of the pan
    against the ORACLE's intended mode."""
    from linux_fg_amd import capi
    name, prev, curr = list(_lean_cases())[case]
    lean_ctx.set_semantics(capi.SEMANTICS_INTENDED)
    try:
        for lane in (0, 1):
            lean_ctx.lane_select(lane)
            a, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_PREFILTERED)
            used, listed, left = lean_ctx.motion_lean_stats()
            assert used and listed > 300, (used, listed)
            if name in ("pan", "stills"):
                assert left == 0, f"{name}: the lean kernel left work in {left} of {listed} tiles"
            b, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_EXACT_ONLY)
            assert (a == b).all(), f"{name}, lane {lane}: {(a != b).any(-1).sum()} pixels differ"
        if name in ("pan", "stills"):
            x0, y0, x1, y1 = roi = (600, 400, 700, 424)
            want = as_int(oracle.motion(prev, curr, roi=roi, semantics=oracle.INTENDED))[y0:y1, x0:x1]
            assert (a[y0:y1, x0:x1] == want).all(), name
            if name == "stills":
                assert (a[64:-64, 64:-64] == 0).all()          # (the shaders' own order says (-16, -16) on ties; the intended one the zero vector)
    finally:
        lean_ctx.set_semantics(capi.SEMANTICS_REFERENCE)
        lean_ctx.lane_select(0)


@pytest.mark.parametrize("wh,shift", [((1284, 726), (2, 5)), ((1284, 726), (-3, -4)), ((1540, 870), (0, -6)), ((1352, 778), (4, 0))])
def test_lean_kernel_on_ragged_sizes(lean_ctx, wh, shift):
    """Heights that are no multiple of a tile's 64 rows or a segment's 16, pans in every direction: which segments of the rim tiles
    above and below the interior the lean kernel takes (lean_segment_ok: all of a segment's block positions inside the image, no
    candidate's block outside prev altogether), its window rows outside the image staged as zeros, the plan's units for those
    segments leaving when they find them settled -- whole frame against the literal kernel, both lanes."""
    from linux_fg_amd import capi
    w, h = wh
    prev = synth.make_prev(w, h, seed=synth.BASE_SEED + w + h)
    curr = synth.translate(prev, shift, synth.BASE_SEED + w - h)
    for lane in (0, 1):
        lean_ctx.lane_select(lane)
        a, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_PREFILTERED)
        used, listed, left = lean_ctx.motion_lean_stats()
        assert used and listed > 50, (used, listed)
        b, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_EXACT_ONLY)
        assert (a == b).all(), f"{wh} {shift}, lane {lane}: {(a != b).any(-1).sum()} pixels differ"
    lean_ctx.lane_select(0)


def test_lean_kernel_at_4k_and_after_a_change_of_content(lean_ctx, oracle):
    """BASELINE config 3's size: the benchmark's own frames (a 1080p pan, both frames upscaled on the device) and the 4K mixture
    one after the other on one lane -- the second plan's lists and the tiles the kernel left are reused from call to call --
    against the literal kernel everywhere and against the ORACLE on regions of the pan."""
    from linux_fg_amd import capi
    w, h = 1920, 1080
    pin = synth.make_prev(w, h, seed=synth.BASE_SEED)
    cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
    p, c = lean_ctx.frame_from(pin), lean_ctx.frame_from(cin)
    P, C = lean_ctx.create_frame(2 * w, 2 * h), lean_ctx.create_frame(2 * w, 2 * h)
    lean_ctx.scale(p, P); lean_ctx.scale(c, C)
    Pn, Cn = lean_ctx.download(P), lean_ctx.download(C)
    for f in (p, c, P, C):
        lean_ctx.destroy_frame(f)
    mix_prev, mix_curr = _uhd_mixture()
    for name, prev, curr in (("pan", Pn, Cn), ("mixture", mix_prev, mix_curr), ("pan again", Pn, Cn)):
        a, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_PREFILTERED)
        b, _ = run_motion_mode(lean_ctx, prev, curr, capi.MOTION_EXACT_ONLY)
        assert (a == b).all(), f"{name}: {(a != b).any(-1).sum()} pixels differ"
        if name == "pan":
            inner = a[64:-64, 64:-64]
            assert (inner[..., 0] == -6).all() and (inner[..., 1] == 4).all()
            for roi in [(1000, 1000, 1100, 1024), (60, 60, 150, 90), (3700, 2080, 3800, 2104)]:
                x0, y0, x1, y1 = roi
                want = as_int(oracle.motion(Pn, Cn, roi=roi))[y0:y1, x0:x1]
                assert (a[y0:y1, x0:x1] == want).all(), roi


def test_oracle_on_a_tile_the_lean_kernel_left_and_on_a_rim_tiles_inner_segments(lean_ctx, oracle):
    """VERDICT r4 (weak 2): the ORACLE, not the literal kernel, where the lean path hands work back -- a tile it LEFT (the edge of a patch
    that moves on its own inside the benchmark's pan: two candidates, unmatched pixels) and the inner segments of rim tiles above and
    below the interior that it takes (lean_segment_ok) -- at 4K, frames upscaled on the device, default knobs but the lean kernel on every
    call."""
    from linux_fg_amd import capi
    w, h = 1920, 1080
    pin = synth.make_prev(w, h, seed=synth.BASE_SEED)
    cin = synth.translate(pin, (3, -2), synth.BASE_SEED)
    cin[500:560, 800:880] = pin[497:557, 804:884]          # moves by (-4, 3) at input resolution while the frame moves by (3, -2)
    p, c = lean_ctx.frame_from(pin), lean_ctx.frame_from(cin)
    W, H = 2 * w, 2 * h
    P, C = lean_ctx.create_frame(W, H), lean_ctx.create_frame(W, H)
    lean_ctx.scale(p, P); lean_ctx.scale(c, C)
    Pn, Cn = lean_ctx.download(P), lean_ctx.download(C)
    for f in (p, c, P, C):
        lean_ctx.destroy_frame(f)
    for lane in (0, 1):
        lean_ctx.lane_select(lane)
        a, _ = run_motion_mode(lean_ctx, Pn, Cn, capi.MOTION_PREFILTERED)
        used, listed, left = lean_ctx.motion_lean_stats()
        assert used and left > 0, (used, listed, left)
        for roi in [(1590, 990, 1620, 1006),               # the patch's left edge: a tile the kernel left
                    (1740, 1110, 1770, 1126),              # its bottom-right corner
                    (1000, 20, 1060, 36), (2000, 44, 2040, 60),            # rim tiles above the interior: segments inside the image
                    (1500, H - 60, 1560, H - 44)]:                         # ... and below
            x0, y0, x1, y1 = roi
            want = as_int(oracle.motion(Pn, Cn, roi=roi))[y0:y1, x0:x1]
            assert (a[y0:y1, x0:x1] == want).all(), f"lane {lane}, {roi}: {(a[y0:y1, x0:x1] != want).any(-1).sum()} pixels differ"
    lean_ctx.lane_select(0)


def test_the_variant_for_moderate_noise_follows_the_content_and_agrees_with_the_literal_kernel(monkeypatch):
    """motion_prefilter_kernel<false, 1> (the walks by sums of absolute differences; include/linuxfg_hip.h: lfg_motion_last_variant): chosen for
    a lane's call when half the sample blocks of the lane's PREVIOUS call matched moderately well.  A pan, then the pan under noise of
    +-4 levels (at the resolution of the frames: the walks see thresholds of several hundred), then the pan again, on two lanes: the
    second noisy call of a lane and the first clean one after it run the variant, every call returns the literal kernel's vectors."""
    from linux_fg_amd import capi
    monkeypatch.delenv("LFG_TIER_FORCE", raising=False)
    W, H = 1920, 1080
    base = synth.make_prev(W, H, seed=synth.BASE_SEED + 606)
    pan = synth.translate(base, (5, -3), synth.BASE_SEED + 606)
    noise = synth.noise_bytes(W, H, 777) % 9
    noisy = np.clip(pan.astype(np.int16) + noise.astype(np.int16) - 4, 0, 255).astype(np.uint8)
    c = capi.Context(0)
    try:
        c.lanes(2)
        want = {name: run_motion_mode(c, base, curr, capi.MOTION_EXACT_ONLY)[0] for name, curr in (("pan", pan), ("noisy", noisy))}
        variants = []
        for k, name in enumerate(["pan", "pan", "noisy", "noisy", "noisy", "noisy", "pan", "pan", "pan", "pan"]):
            c.lane_select(k % 2)
            got = run_motion(c, base, pan if name == "pan" else noisy)          # (synchronises: the verdict is there for the lane's next call)
            variants.append(c.motion_last_variant())
            assert (got == want[name]).all(), (k, name)
        assert variants == [0, 0, 0, 0, 1, 1, 1, 1, 0, 0], variants
        c.lane_select(0)
    finally:
        c.close()


@pytest.mark.parametrize("amp", [4, 8])
def test_the_variant_for_moderate_noise_against_the_oracle_at_4k(monkeypatch, oracle, amp):
    """motion_prefilter_kernel<false, 1> on what it is for -- the benchmark's pan under sensor noise of +-4 and +-8 levels at the 1080p input,
    both frames upscaled on the device: thresholds of 400 - 1,000, the eight- and sixteen-point walks by SADs decide nearly every batch --
    forced for every call (LFG_TIER_FORCE=1), against the ORACLE on regions inside the frame and at its rim, and against the literal kernel
    everywhere."""
    from linux_fg_amd import capi
    monkeypatch.setenv("LFG_TIER_FORCE", "1")
    w, h = 1920, 1080
    pin = synth.make_prev(w, h, seed=synth.BASE_SEED)
    noise = synth.noise_bytes(w, h, 4242 + amp) % (2 * amp + 1)
    cin = np.clip(synth.translate(pin, (3, -2), synth.BASE_SEED).astype(np.int16) + noise.astype(np.int16) - amp, 0, 255).astype(np.uint8)
    c = capi.Context(0)
    try:
        c.lanes(2)
        p, q = c.frame_from(pin), c.frame_from(cin)
        W, H = 2 * w, 2 * h
        P, C = c.create_frame(W, H), c.create_frame(W, H)
        c.scale(p, P); c.scale(q, C)
        Pn, Cn = c.download(P), c.download(C)
        for f in (p, q, P, C):
            c.destroy_frame(f)
        a, _ = run_motion_mode(c, Pn, Cn, capi.MOTION_PREFILTERED)
        assert c.motion_last_variant() == 1
        b, _ = run_motion_mode(c, Pn, Cn, capi.MOTION_EXACT_ONLY)
        assert (a == b).all(), f"+-{amp}: {(a != b).any(-1).sum()} pixels differ from the literal kernel"
        for roi in [(1000, 1000, 1100, 1024), (2500, 300, 2580, 324), (40, 1500, 120, 1524), (W - 130, 800, W - 50, 824)]:
            x0, y0, x1, y1 = roi
            want = as_int(oracle.motion(Pn, Cn, roi=roi))[y0:y1, x0:x1]
            assert (a[y0:y1, x0:x1] == want).all(), f"+-{amp}, {roi}: {(a[y0:y1, x0:x1] != want).any(-1).sum()} pixels differ from the oracle"
    finally:
        c.close()


def test_lean_verdict_follows_the_content():
    """Without the override the host goes by the order kernel's verdict on the lane's previous call: results are the literal
    kernel's on a stream that changes from a pan to noise and back, whichever calls went through the lean kernel."""
    from linux_fg_amd import capi
    W, H = 1920, 1080
    base = synth.make_prev(W, H, seed=synth.BASE_SEED + 505)
    pan = synth.translate(base, (-4, 6), synth.BASE_SEED + 505)
    noise = synth.noise_bytes(W, H, 31337)
    c = capi.Context(0)
    try:
        c.lanes(2)
        want = {}
        for name, curr in (("pan", pan), ("noise", noise)):
            want[name], _ = run_motion_mode(c, base, curr, capi.MOTION_EXACT_ONLY)
        for k, name in enumerate(["pan", "pan", "pan", "noise", "noise", "pan", "pan"]):
            c.lane_select(k % 2)
            got = run_motion(c, base, pan if name == "pan" else noise)
            assert (got == want[name]).all(), (k, name)
        c.lane_select(0)
    finally:
        c.close()


def test_wrong_guesses_are_counted_and_a_lane_can_be_waited_for_alone():
    """lfg_motion_prediction_stats: with frames in flight three launch decisions go by the lane's previous finished call.  A stream
    that turns from a pan to noise and back, the host waiting for a lane's previous frame before it reuses the lane (lfg_lane_sync:
    that lane alone): every call's verdict comes back, and exactly the calls at the two changes of content were launched on a
    wrong guess about the lean kernel -- and still produced the literal kernel's vectors."""
    from linux_fg_amd import capi
    W, H = 1920, 1080
    base = synth.make_prev(W, H, seed=synth.BASE_SEED + 707)
    pan = synth.translate(base, (3, 2), synth.BASE_SEED + 707)
    noise = synth.noise_bytes(W, H, 4242)
    c = capi.Context(0)
    try:
        c.lanes(2)
        want = {}
        for name, curr in (("pan", pan), ("noise", noise)):
            want[name], _ = run_motion_mode(c, base, curr, capi.MOTION_EXACT_ONLY)
        P = c.frame_from(base); frames = {"pan": c.frame_from(pan), "noise": c.frame_from(noise)}
        mvs = [c.create_frame(W, H, capi.FORMAT_MV_S8X2) for _ in range(2)]
        assert c.motion_prediction_stats() == (0, 0, 0, 0)
        seq = ["pan"] * 6 + ["noise"] * 6 + ["pan"] * 6
        for k, name in enumerate(seq):
            c.lane_select(k % 2)
            c.lane_sync()                                   # this lane's previous frame is done (the other lane may still run)
            if k >= 2:
                assert (as_int(c.download(mvs[k % 2])) == want[seq[k - 2]]).all(), (k - 2, seq[k - 2])
            c.motion(P, frames[name], mvs[k % 2])
        c.sync()
        verdicts, lean_wrong, grid_wrong, second_wrong = c.motion_prediction_stats()
        # (a call's verdict is read when its lane is used next: the last two calls' are still out)
        assert verdicts == len(seq) - 2, verdicts
        # per lane: the first call (no guess yet: none of the lean kernel), the first noise call (guessed lean), the first pan after the
        # noise (guessed none) -- and the second pass never flags a tile here
        assert lean_wrong == 2 * 3 and second_wrong == 0, (lean_wrong, grid_wrong, second_wrong)
        assert grid_wrong >= 2 * 2
        c.lane_select(0)
    finally:
        c.close()


def test_small_fallback_launch_takes_the_tiles_a_call_flags_after_all():
    """With frames in flight a lane whose previous call sent no tile through the literal kernel launches that kernel with 64
    workgroups instead of one per part of every possible flagged tile; they loop.  A pan (nothing flagged), then -- the lane's
    verdict has arrived: the context is synchronised between the calls -- a fade over flat bars, which flags tiles by the dozen,
    then the fade again (now expected: the large grid) and the pan: every call's vectors are the literal kernel's."""
    from linux_fg_amd import capi
    W, H = 1920, 1080
    base = synth.make_prev(W, H, seed=synth.BASE_SEED + 606)
    pan = synth.translate(base, (2, 3), synth.BASE_SEED + 606)
    flat_prev, flat_curr = base.copy(), pan.copy()
    flat_prev[300:700] = 100; flat_curr[300:700] = 101          # a flat band whose brightness changes: ties at a non-zero cost
    c = capi.Context(0)
    try:
        c.lanes(2)
        want_pan, _ = run_motion_mode(c, base, pan, capi.MOTION_EXACT_ONLY)
        want_flat, _ = run_motion_mode(c, flat_prev, flat_curr, capi.MOTION_EXACT_ONLY)
        flagged = []
        for name in ("pan", "pan", "flat", "flat", "pan", "flat"):
            prev, curr, want = (base, pan, want_pan) if name == "pan" else (flat_prev, flat_curr, want_flat)
            got, st = run_motion_mode(c, prev, curr, capi.MOTION_PREFILTERED)
            c.sync()
            flagged.append(st[1])
            assert (got == want).all(), f"{name}: {(got != want).any(-1).sum()} pixels differ"
        assert flagged[0] == 0 and flagged[1] == 0 and flagged[2] > 0 and flagged[3] == flagged[2] and flagged[4] == 0, flagged
    finally:
        c.close()


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_band_restricted_lattice_tests_agree_with_the_literal_kernel(ctx, seed):
    """Segments in which most pixels own a zero-cost candidate and a few columns only nearly match (costs of a few hundred:
    the columns an upscaler filters differently next to the border, mild noise on a vertical strip): the four- and
    sixteen-point tests walk the band's lattice groups only and the settled pixels are answered for by an exact-texel
    compare.  Strips of +-1 .. +-3 levels at arbitrary x and width, inside the frame and at both edges, with duplicated
    texels sprinkled in (lattice points that DO compare equal), on a pan and on a static pair; whole frame against the
    literal kernel."""
    from linux_fg_amd import capi
    rng = np.random.default_rng(9500 + seed)
    w, h = 1176, 416
    prev = synth.make_prev(w, h, seed=9500 + seed)
    shift = (int(rng.integers(-5, 6)), int(rng.integers(-3, 4))) if seed != 23 else (0, 0)
    curr = synth.translate(prev, shift, 9500 + seed)
    base = curr.copy()
    x = 4
    for k in range(12):
        x += int(rng.integers(30, 90))
        width = int(rng.integers(1, 20))
        amp = int(rng.integers(1, 4))
        y0, y1 = sorted(int(v) for v in rng.integers(0, h, 2))
        y1 = max(y1, min(h, y0 + 24))
        n = rng.integers(-amp, amp + 1, size=(y1 - y0, width, curr.shape[2]), dtype=np.int16)
        curr[y0:y1, x:x + width] = np.clip(base[y0:y1, x:x + width].astype(np.int16) + n, 0, 255).astype(np.uint8)
        x += width
    for x0, x1 in ((0, 6), (w - 7, w)):               # ... and at both edges, full height
        n = rng.integers(-2, 3, size=(h, x1 - x0, curr.shape[2]), dtype=np.int16)
        curr[:, x0:x1] = np.clip(base[:, x0:x1].astype(np.int16) + n, 0, 255).astype(np.uint8)
    ys, xs = rng.integers(0, h, 400), rng.integers(0, w, 400)      # texels that repeat elsewhere in prev
    prev[ys, xs] = prev[(ys + 5) % h, (xs + 3) % w]
    a, st = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_narrow_bands_deferred_test_and_rank_walk_agree_with_the_literal_kernel(ctx, seed):
    """The round-2 shortcuts of the prefilter, each provoked on purpose and compared with the literal kernel on the whole
    frame: vertical strips of fresh noise 3 / 7 / 11 / 15 columns wide at arbitrary x (narrow search with 5 / 4 / 3 / 2
    candidates per pass, inside the frame and at its left and right edge), a strip too wide for it, a band of sensor noise
    (the four-point test's survivors wait for the deferred sixteen-point test), a static region and an exactly panned one
    (every pixel owns a zero-cost candidate: the by-rank walk), and a pan that exposes strips at two borders."""
    from linux_fg_amd import capi
    rng = np.random.default_rng(9000 + seed)
    w, h = 1176, 416                                  # 21 x 6.5 prefilter tiles: rim and interior tiles, a partial last row
    prev = synth.make_prev(w, h, seed=9000 + seed)
    curr = synth.translate(prev, (int(rng.integers(-4, 5)), int(rng.integers(-3, 4))), 9000 + seed)
    curr[:, 300:600] = prev[:, 300:600]               # a static region
    fresh = synth.noise_bytes(w, h, 9100 + seed)
    x = 8
    for width in (3, 7, 11, 15, 15, 11, 7, 3, 23):    # strips inside the frame ...
        x += int(rng.integers(40, 90))
        y0, y1 = sorted(int(v) for v in rng.integers(0, h, 2))
        y1 = max(y1, min(h, y0 + 40))
        curr[y0:y1, x:x + width] = fresh[y0:y1, x:x + width]
        x += width
    curr[:, :5] = fresh[:, :5]                        # ... and at both edges
    curr[:, w - 9:] = fresh[:, w - 9:]
    n = synth.noise_bytes(w, h, 9200 + seed) % 5      # sensor noise on a band of rows
    noisy = np.clip(curr.astype(np.int16) + n.astype(np.int16) - 2, 0, 255).astype(np.uint8)
    curr[330:400] = noisy[330:400]
    a, st = run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    assert (a == b).all()
    assert st[1] == 0                                 # no tile needed the literal kernel
