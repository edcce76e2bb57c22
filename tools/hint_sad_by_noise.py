"""Best-match SAD of 8 x 8 sample blocks (what motion_hint_kernel reports per block) on the benchmark's pan under sensor noise of
+-1 .. +-12 levels at the 1080p input, frames upscaled on the device: where the order kernel's "moderate" verdict has to draw its line."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from linux_fg_amd import capi

w_in, h_in = 1920, 1080
with capi.Context(0) as ctx:
    for amp in (1, 2, 3, 4, 6, 8, 12):
        os.environ["LFG_BENCH_NOISE_AMP"] = str(amp)
        p_in, c_in = bench.make_content("noisy", w_in, h_in, 0, 0)
        fp, fc = ctx.frame_from(p_in), ctx.frame_from(c_in)
        P, C = ctx.create_frame(2 * w_in, 2 * h_in), ctx.create_frame(2 * w_in, 2 * h_in)
        ctx.scale(fp, P); ctx.scale(fc, C)
        Pn, Cn = ctx.download(P).astype(np.int16), ctx.download(C).astype(np.int16)
        for f in (fp, fc, P, C):
            ctx.destroy_frame(f)
        dx, dy = 6, -4                                  # rank 0's translation (3, -2) at 4K: curr(q) = prev(q - shift)
        sads = []
        rng = np.random.default_rng(1)
        for _ in range(1024):
            x, y = int(rng.integers(40, 3840 - 48)), int(rng.integers(40, 2160 - 48))
            sads.append(int(np.abs(Cn[y:y + 8, x:x + 8] - Pn[y - dy:y - dy + 8, x - dx:x - dx + 8]).sum()))
        q = np.percentile(sads, [5, 25, 50, 75, 95])
        print("noise +-%2d: SAD of the true candidate, 1024 blocks: 5 %% %5.0f  25 %% %5.0f  median %5.0f  75 %% %5.0f  95 %% %5.0f" % (amp, *q), flush=True)
