"""what does the small second pass cost when content changes under frames in flight?  A stream of pan frames, then
a cut to flat frames (a fade: every tile of every call goes through the literal kernel), three frames in flight, paced as a host that
reuses a lane only when its previous frame is done.  Prints each step's completion time relative to the previous completion."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from linux_fg_amd import capi


def run(lanes=3, pans=9, fades=9):
    dev = torch.device("cuda", 0)
    w_in, h_in = bench.SIZES["1080p"]
    w, h = 2 * w_in, 2 * h_in
    ctx = capi.Context(0)
    ctx.lanes(lanes)
    def frame(width, height, fmt=capi.FORMAT_RGBA8):
        t = torch.empty((height, width, 4 if fmt == capi.FORMAT_RGBA8 else 2), dtype=torch.uint8, device=dev)
        return t, capi.Context.wrap(t.data_ptr(), width, height, fmt)
    contents = {}
    for name in ("translated", "fade"):
        p_in, c_in = bench.make_content(name, w_in, h_in, 0, 0)
        tp, fp = frame(w_in, h_in); tc, fc = frame(w_in, h_in)
        tp.copy_(torch.from_numpy(p_in)); tc.copy_(torch.from_numpy(c_in))
        tp4, fp4 = frame(w, h)
        ctx.scale(fp, fp4)
        contents[name] = (tp, tc, tp4, fp4, fc)
    bufs = [(frame(w, h), frame(w, h, capi.FORMAT_MV_S8X2), frame(w, h)) for _ in range(lanes)]
    ctx.sync(); torch.cuda.synchronize()
    order = ["translated"] * pans + ["fade"] * fades + ["translated"] * pans
    def step(k, name):
        j = k % lanes
        (_, fc4), (_, fmv), (_, fout) = bufs[j]
        _, _, _, fp4, fc = contents[name]
        ctx.lane_select(j)
        ctx.lane_wait((k - 1) % lanes)
        ctx.scale(fc, fc4)
        ctx.lane_mark()
        ctx.motion(fp4, fc4, fmv, 8, 16.0)
        ctx.interpolate(fp4, fc4, fmv, fout, 0.5)
    for k in range(4 * lanes):                       # warm: workspaces, verdicts
        step(k, "translated")
    ctx.sync()
    done = {}
    t0 = time.perf_counter()
    for k, name in enumerate(order):
        if k >= lanes:
            ctx.lane_select(k % lanes); ctx.lane_sync()
            done[k - lanes] = time.perf_counter() - t0
        step(k, name)
    for k in range(len(order) - lanes, len(order)):
        ctx.lane_select(k % lanes); ctx.lane_sync()
        done[k] = time.perf_counter() - t0
    ctx.lane_select(0)
    prev = 0.0
    for k, name in enumerate(order):
        print("step %2d %-10s done at %8.3f ms  (+%7.3f)" % (k, name, done[k] * 1e3, (done[k] - prev) * 1e3))
        prev = done[k]
    print("verdicts read back / lean guessed wrong / grid guessed wrong / small second pass met flagged tiles:", ctx.motion_prediction_stats())
    ctx.close()


if __name__ == "__main__":
    for env in ({}, {"LFG_FALLBACK_FULL": "1"}):
        os.environ.pop("LFG_FALLBACK_FULL", None)
        os.environ.update(env)
        print("== second pass:", "always its full grid (LFG_FALLBACK_FULL=1)" if env else "256 workgroups while the lane's last call flagged no tile (default)")
        run()
