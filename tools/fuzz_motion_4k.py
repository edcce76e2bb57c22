"""On the GPU box: extended seeded fuzz of the prefiltered motion path against the literal kernel at 4K-class sizes
(48 content mixtures from tests/test_gpu_parity.py::_mixed_pair, every fourth under the intended tie order).
Not part of the test suite (48 cases take ~20 s; LFG_FUZZ_CASES=N for more); last run (round 2: narrow search, deferred
sixteen-point test, by-rank walk, band-restricted tests, inherited thresholds, list depths 32/24/24): 0 differences in
320 cases; round 3 (lists of depth 10 with the restart rule, the resolve kernel on the list of open segments, the
single fallback launch): 0 differences in 200 cases one frame at a time and 120 with LFG_FUZZ_LANES=3 (the plan of a
context with frames in flight); the round's final library (lattice walks in four copies, survivors always deferred, ranks by
arithmetic, thresholds written late, kernels templated for the north-star order): 0 differences in 400 + 240 cases; round 4's
final library (lean kernel incl. the rim tiles' inner segments, row band, eight-point and SAD four-point walks, heads of two
entries, handed-over segments in four parts with frames in flight): 0 differences in 300 cases one frame at a time, 200 with
LFG_FUZZ_LANES=3 and 160 with LFG_FUZZ_LANES=3 LFG_LEAN_FORCE=1 (every call through the lean kernel); and again behind the
launches sized by the lane's previous call (persistent grid, resolve grid, the looping fallback pass): 200 with LFG_FUZZ_LANES=3,
120 without, no differences; round 5 (the variant of the persistent kernel for moderate noise forced, LFG_TIER_FORCE=1, noise amplitudes
up to 12 levels, LFG_FUZZ_MAX_AMP=12; the visiting order dealt out by LDS bank): 384 cases, see NOTES_r05.md section 6; round 5's final library
(tools/gpu_r5_fuzz_final.sh): 300 cases one frame at a time, 200 with LFG_FUZZ_LANES=3, 160 more with LFG_LEAN_FORCE=1 and 160 with
LFG_MOTION_STRIP=1 (the strip kernel), no differences."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
from linux_fg_amd import capi, synth
import test_gpu_parity as T
ctx = capi.Context(0)
if int(os.environ.get("LFG_FUZZ_LANES", "1")) > 1:      # (the plan of a context with frames in flight: rim segments in four parts)
    ctx.lanes(int(os.environ["LFG_FUZZ_LANES"]))
bad = 0
for case in range(int(os.environ.get("LFG_FUZZ_CASES", "48"))):
    rng = np.random.default_rng(77000 + case)
    w, h = int(rng.integers(2500, 3900)), int(rng.integers(1400, 2200))
    prev, curr = T._mixed_pair(w, h, 77000 + case, int(os.environ.get("LFG_FUZZ_MAX_AMP", "4")))    # (12: costs of up to a thousand -- the walks by SADs of the persistent kernel's variant, LFG_TIER_FORCE=1)
    if case % 4 == 3: ctx.set_semantics(capi.SEMANTICS_INTENDED)
    a, st = T.run_motion_mode(ctx, prev, curr, capi.MOTION_PREFILTERED)
    b, _ = T.run_motion_mode(ctx, prev, curr, capi.MOTION_EXACT_ONLY)
    ctx.set_semantics(capi.SEMANTICS_REFERENCE)
    d = int((a != b).any(-1).sum())
    bad += d != 0
    print(case, w, h, "differ", d, "fallback tiles", st[1], flush=True)
print("cases with differences:", bad)
