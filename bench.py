#!/usr/bin/env python3
"""Benchmark of the MI355X-native linux-fg hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload pipeline|scale]

Metric (BASELINE.json): interpolated frames/s at 1080p -> 4K RGBA8, with the achieved rate of the
dominant kernel against its roofline.  A "step" is one pass of the hot path over one synthetic
frame pair per GPU, inputs already resident in HBM:

  pipeline (default; BASELINE config 3, north_star order)
      scale(curr 1080p -> 4K)  ->  motion(prev4K, curr4K; blockSize 8, searchRadius 16)
      ->  interpolate(prev4K, curr4K, mv, t = 0.5)          = one interpolated 4K frame
  scale (BASELINE config 2): the Lanczos kernel alone, 1080p -> 4K, one upscaled frame per step.
  pipeline_input_res (labelled variant): the reference's own data flow keeps prev/curr at input
      resolution (src/scaler.cpp:443,451), so motion + interpolate run at 1080p and both the real and the
      generated frame are upscaled: motion(prev, curr) -> interpolate -> scale(curr) + scale(interpolated).
      Same deliverables per step (one real and one generated 4K frame) but NOT the same pixels as the
      north_star order; never the headline value.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), launched by torch.distributed.run.
Frame pairs are independent, so the work shards one pair per GPU with no data-path collective except
the one the path really has: the batch's shared previous frame is broadcast from rank 0 each step
(double-buffered, issued one step ahead so it overlaps the kernels) -- as the 8.3 MB input frame, which
every rank upscales itself (one more 17 us scale per step than at N = 1; 33 MB of upscaled frame per
0.8 ms step would cost more on xGMI than that).  Weak scaling.

The CPU baseline is the oracle (oracle/lfg_oracle.c, a restatement of the reference shaders -- NOT
lavapipe, which this image lacks) timed on a bounded sample on the host cores, rank 0, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md:36 (spec; 6290 measured copy)
FP32_VALU_PEAK_TFLOPS = 157.3  # ibid. :41 (counts an FMA as 2; an add-only stream tops out at 78.65)

SIZES = {"540p": (960, 540), "1080p": (1920, 1080), "4k": (3840, 2160), "8k": (7680, 4320)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["pipeline", "scale", "pipeline_input_res"], default="pipeline")
    ap.add_argument("--input", choices=list(SIZES), default="1080p", help="input size; output is 2x")
    ap.add_argument("--factors", default="0.5", help="comma-separated interpolation factors per pair")
    ap.add_argument("--content", choices=["translated", "occluded", "objects", "noisy", "uncorrelated", "static", "fade"], default="translated",
                    help="translated (default: curr = prev shifted by (3,-2)), occluded (the same with patches of fresh noise), objects (the same with patches that move on their own), noisy (the same with +-2 levels of noise everywhere), uncorrelated (independent noise frames), "
                         "static (curr = prev), fade (flat grey frames one level apart: every candidate ties at a "
                         "non-zero cost, the prefilter's worst case -- all tiles fall back to the literal kernel)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def algorithmic_bytes(stage: str, w_in: int, h_in: int, w: int, h: int) -> int:
    """every stage reads each input once and writes each output once;
    frames 4 B/px, motion vectors 2 B/px."""
    if stage == "scale":
        return 4 * (w_in * h_in + w * h)
    if stage == "motion":
        return 8 * w * h + 2 * w * h
    if stage == "interpolate":
        return 8 * w * h + 2 * w * h + 4 * w * h
    raise ValueError(stage)


PREFILTER_LAUNCHES_PER_CALL = 2          # motion_prefilter_kernel: the plan's units, then the queue of segments handed over


def pmc_traffic(kernel_prefixes):
    """HBM bytes per launch of the kernels whose names start with one of `kernel_prefixes`, summed, from the
    newest committed two-pass PMC summary (profiles/rNN_hbm_traffic_pmc.txt: FETCH_SIZE and WRITE_SIZE
    collected in separate rocprofv3 passes, values in KB).  bench.py cannot run the profiler on itself;
    None if no summary exists."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.txt")))
    if not files:
        return None, None
    if isinstance(kernel_prefixes, str):
        kernel_prefixes = [kernel_prefixes]
    vals, cur = {}, None
    for line in open(files[-1]):
        if ": launches" in line and not line.startswith(" "):        # any kernel header line starts a new block
            cur = line.split(": launches")[0].replace("void ", "").split("<")[0].split("(")[0].strip()
        m = re.match(r"^\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.e+]+)", line)
        if m and cur and any(cur.startswith(k) for k in kernel_prefixes):
            # (the summary averages over launches; lfg_motion launches the prefilter twice: the tiles, then the segments
            #  handed over -- an almost empty launch on the benchmark frames -- so a call is two launches)
            per_call = PREFILTER_LAUNCHES_PER_CALL if cur.startswith("lfg::motion_prefilter") else 1
            vals[(cur, m.group(1))] = float(m.group(2)) * 1024.0 * per_call       # the last block of a kernel wins
    have = {k for k, _ in vals}
    if have and all((k, c) in vals for k in have for c in ("FETCH_SIZE", "WRITE_SIZE")):
        return int(sum(vals.values())), os.path.relpath(files[-1], ROOT)
    return None, None


def pmc_executed(kernel_prefix):
    """What the dominant kernel actually executed per launch, from the newest committed SQ counter summary
    (profiles/rNN_motion_sq_counters.txt, one rocprofv3 --pmc pass over tools/run_stage.py motion): wave-level
    VALU / LDS / SALU instruction counts and the kernel's mean duration in that pass.  None if absent."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_motion_sq_counters.txt")))
    if not files:
        return None
    out, cur = {}, None
    for line in open(files[-1]):
        if ": launches" in line and not line.startswith(" "):
            cur = line.split(": launches")[0].replace("void ", "").split("<")[0].split("(")[0].strip()
            m = re.search(r"mean ([0-9.]+) us", line)
            if cur.startswith(kernel_prefix) and m:
                out = {"mean_us": float(m.group(1))}
        m = re.match(r"^\s+(SQ_\w+)\s+([0-9.e+]+)", line)
        if m and cur and cur.startswith(kernel_prefix):
            out[m.group(1)] = float(m.group(2))
    if not {"mean_us", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"} <= set(out):
        return None
    simds, clock_hz = 1024, 2.4e9                       # 256 CUs x 4 SIMDs, peak engine clock
    n = PREFILTER_LAUNCHES_PER_CALL if kernel_prefix.startswith("lfg::motion_prefilter") else 1   # the summary averages over launches
    slots = out["mean_us"] * 1e-6 * clock_hz * simds
    return {"source": os.path.relpath(files[-1], ROOT), "kernel_us_per_call_in_that_pass": round(out["mean_us"] * n, 2),
            "launches_per_call": n,
            "valu_wave_instructions": out["SQ_INSTS_VALU"] * n, "lds_wave_instructions": out["SQ_INSTS_LDS"] * n,
            "salu_wave_instructions": out["SQ_INSTS_SALU"] * n,
            "valu_issue_utilisation": round(out["SQ_INSTS_VALU"] * 4.0 / slots, 3),
            "how": "VALU wave-instructions x 4 cycles / (duration x 2.4 GHz x 1024 SIMDs): the share of the VALU issue "
                   "slots the kernel filled -- the utilisation figure that `frac` (an algorithmic rate) is not"}


def rank_motion(rank: int):
    """The translation (input pixels) of rank `rank`'s current frame against the shared previous frame.  It has to stay
    inside the motion search range after the 2x upscale (|2 dx|, |2 dy| <= 16), or that rank's frames have no match
    anywhere and its motion stage searches in full: (3, -2) on rank 0, then (4, -2) ... (7, -2), (3, -3) ..."""
    return (3 + rank % 5, -2 - (rank // 5) % 6)


def motion_flops(w: int, h: int, block: int = 8, radius: int = 16) -> float:
    """Algorithmic flops of motion.comp with per-position distance reuse: per candidate, one distance
    per pixel (4 sub, 4 mul, 3 add, 1 sqrt = 12 flops) and block*block adds per pixel."""
    cand = (2 * radius + 1) ** 2
    return float(w) * h * cand * (block * block + 12)


def cpu_baseline(w_in, h_in, w, h, factors, workload):
    """Oracle timed on a bounded sample (~10-20 s) of the same workload on the host cores."""
    import oracle
    from linux_fg_amd import synth
    threads = oracle.default_threads()
    prev_in, curr_in = synth.make_pair(w_in, h_in, stream=0)
    # scale: a band of output rows, full width
    band = h                                                          # the whole frame: ~1 s
    t0 = time.perf_counter()
    oracle.scale(curr_in, w, h, roi=(0, 0, w, band), threads=threads)
    t_scale = (time.perf_counter() - t0) * (h / band)
    if workload == "scale":
        return {"value": 1.0 / t_scale, "unit": "upscaled frames/s", "cores": threads, "kind": "port",
                "sample": f"oracle scale.comp restatement on {band} of {h} output rows at {w_in}x{h_in}->{w}x{h}, "
                          f"extrapolated by rows; {threads} threads"}
    # motion + interpolate at output resolution (input resolution for the labelled variant) on synthetic frames
    n_scales = 1
    if workload == "pipeline_input_res":
        w, h = w_in, h_in
        n_scales = 1 + len(factors)
    prev, curr = synth.make_pair(w, h, stream=0)
    mw, mh = min(512, w // 2), min(512, h // 2)                      # crop away from the borders: ~6 s at 16 threads
    t0 = time.perf_counter()
    x0, y0 = (w - mw) // 2, (h - mh) // 2
    mv_roi = oracle.motion(prev, curr, roi=(x0, y0, x0 + mw, y0 + mh), threads=threads)
    t_motion = (time.perf_counter() - t0) * (w * h / (mw * mh))
    iband = h
    t0 = time.perf_counter()
    oracle.interpolate(prev, curr, mv_roi, 0.5, roi=(0, 0, w, iband), threads=threads)
    t_interp = (time.perf_counter() - t0) * (h / iband)
    t_pair = n_scales * t_scale + t_motion + len(factors) * t_interp
    return {"value": len(factors) / t_pair, "unit": "interpolated frames/s", "cores": threads, "kind": "port",
            "sample": (f"oracle (CPU restatement of the reference shaders, not lavapipe): {n_scales} x scale on {band}/{2 * h_in} rows, "
                       f"motion on a {mw}x{mh}-pixel crop of the {w}x{h} frame, interpolate on {iband}/{h} rows, "
                       f"each extrapolated by area; {threads} threads; per-frame seconds scale/motion/interpolate = "
                       f"{t_scale:.3f}/{t_motion:.1f}/{t_interp:.3f}")}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # One rank per GPU.  LFG_BENCH_BACKEND=gloo (rehearsal only) lets several ranks share one card so the
    # N > 1 code path can be exercised on a one-GPU box; RCCL itself needs one device per rank.
    backend = os.environ.get("LFG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from linux_fg_amd import capi, sharding, synth

    w_in, h_in = SIZES[args.input]
    w, h = 2 * w_in, 2 * h_in
    factors = [float(x) for x in args.factors.split(",") if x]
    steps = args.steps if args.steps is not None else (200 if args.workload == "scale" else 20)
    warmup = args.warmup if args.warmup is not None else (20 if args.workload == "scale" else 3)

    ctx = capi.Context(dev_index)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    def dev_frame(host: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        return t, capi.Context.wrap(t.data_ptr(), host.shape[1], host.shape[0], capi.FORMAT_RGBA8)

    def empty_frame(width, height, fmt=capi.FORMAT_RGBA8):
        ch = 4 if fmt == capi.FORMAT_RGBA8 else 2
        t = torch.empty((height, width, ch), dtype=torch.uint8, device=dev)
        return t, capi.Context.wrap(t.data_ptr(), width, height, fmt)

    # Synthetic inputs, uploaded once before timing.  The previous frame is shared by all ranks
    # (seed of stream 0); each rank's current frame is that frame translated by its own vector.
    prev_in = synth.make_prev(w_in, h_in, synth.BASE_SEED)
    # Every rank's own motion, inside the search range after the 2x upscale (|2 dx|, |2 dy| <= 16): (3, -2) on rank 0,
    # then (4, -2) ... (7, -2), (3, -3) ...
    content_rank = int(os.environ.get("LFG_BENCH_CONTENT_RANK", rank))       # (diagnostic: another rank's frames on this GPU)
    rank_shift = rank_motion(content_rank)
    if args.content == "translated":
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
    elif args.content == "uncorrelated":
        curr_in = synth.noise_bytes(w_in, h_in, (synth.BASE_SEED + 7919 * (rank + 1)) & 0xFFFFFFFF)
    elif args.content == "occluded":                       # the translated pair with 24 patches of fresh noise (2 % of the frame)
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
        fresh = synth.noise_bytes(w_in, h_in, (synth.BASE_SEED + 104729 * (rank + 1)) & 0xFFFFFFFF)
        rng = np.random.default_rng(20240 + rank)
        for _ in range(24):
            pw, ph = int(rng.integers(w_in // 60, w_in // 12)), int(rng.integers(h_in // 60, h_in // 12))
            x0, y0 = int(rng.integers(40, w_in - 40 - pw)), int(rng.integers(40, h_in - 40 - ph))
            curr_in[y0:y0 + ph, x0:x0 + pw] = fresh[y0:y0 + ph, x0:x0 + pw]
    elif args.content == "objects":                        # the translated pair with 24 patches that move on their own
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
        rng = np.random.default_rng(30240 + rank)
        for _ in range(24):
            pw, ph = int(rng.integers(w_in // 60, w_in // 12)), int(rng.integers(h_in // 60, h_in // 12))
            x0, y0 = int(rng.integers(40, w_in - 40 - pw)), int(rng.integers(40, h_in - 40 - ph))
            dx, dy = int(rng.integers(-7, 8)), int(rng.integers(-7, 8))
            curr_in[y0:y0 + ph, x0:x0 + pw] = prev_in[y0 - dy:y0 - dy + ph, x0 - dx:x0 - dx + pw]
    elif args.content == "noisy":                          # the translated pair plus sensor-like noise: +-2 levels per channel
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
        n = synth.noise_bytes(w_in, h_in, (synth.BASE_SEED + 15485863 * (rank + 1)) & 0xFFFFFFFF) % 5
        curr_in = np.clip(curr_in.astype(np.int16) + n.astype(np.int16) - 2, 0, 255).astype(np.uint8)
    elif args.content == "static":
        curr_in = prev_in.copy()
    else:                                                   # fade
        prev_in = np.full((h_in, w_in, 4), 100, np.uint8)
        curr_in = np.full((h_in, w_in, 4), 101, np.uint8)
    t_prev_in, f_prev_in = dev_frame(prev_in)
    t_curr_in, f_curr_in = dev_frame(curr_in)
    t_curr4, f_curr4 = empty_frame(w, h)
    t_out, f_out = empty_frame(w, h)
    t_mv, f_mv = empty_frame(w, h, capi.FORMAT_MV_S8X2)
    if args.workload == "pipeline_input_res":
        t_mv_in, f_mv_in = empty_frame(w_in, h_in, capi.FORMAT_MV_S8X2)
        t_mid_in, f_mid_in = empty_frame(w_in, h_in)
    # The previous frame.  One GPU: upscaled once, before the timed region (in a stream it is the last step's current
    # frame).  Several GPUs: the batch shares its previous frame, which travels as the 8.3 MB INPUT frame -- one
    # broadcast per step, double-buffered -- and every rank upscales it (17 us) rather than 33 MB of upscaled frame
    # crossing xGMI under a 0.8 ms step.  LFG_BENCH_SHARE_INPUT=1 runs that data flow on one GPU (no collective).
    share_input = world > 1 or os.environ.get("LFG_BENCH_SHARE_INPUT") == "1"
    t_prev4, f_prev4 = empty_frame(w, h)
    ctx.scale(f_prev_in, f_prev4)
    prev_slots = [dev_frame(prev_in) for _ in range(2)] if share_input else [(t_prev4, f_prev4)]
    torch.cuda.synchronize(dev)

    shared_prev = sharding.SharedFrameBroadcaster([t for t, _ in prev_slots], src=0, dist=dist if world > 1 else None,
                                                  world_size=world)

    def step(k):
        # the shared previous frame of this step (waits for its RCCL broadcast, issues the next one)
        t_shared = shared_prev.acquire(k)
        f_shared = prev_slots[[t.data_ptr() for t, _ in prev_slots].index(t_shared.data_ptr())][1]
        f_prev_step = f_shared if share_input else f_prev_in
        if share_input and args.workload == "pipeline":
            ctx.scale(f_shared, f_prev4)
        if args.workload == "pipeline_input_res":
            ctx.motion(f_prev_step, f_curr_in, f_mv_in, 8, 16.0)
            ctx.scale(f_curr_in, f_curr4)
            for t in factors:
                ctx.interpolate(f_prev_step, f_curr_in, f_mv_in, f_mid_in, t)
                ctx.scale(f_mid_in, f_out)
            return
        ctx.scale(f_curr_in, f_curr4)
        if args.workload == "pipeline":
            ctx.motion(f_prev4, f_curr4, f_mv, 8, 16.0)
            for t in factors:
                ctx.interpolate(f_prev4, f_curr4, f_mv, f_out, t)

    def barrier_sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    shared_prev.start(0)
    for k in range(warmup):
        step(k)
    ctx.profile_reset()
    ctx.profile_enable(args.workload != "scale")      # events around every stage launch (ms-scale kernels)
    barrier_sync()
    t0 = time.perf_counter()
    for k in range(warmup, warmup + steps):
        step(k)
    barrier_sync()
    elapsed = time.perf_counter() - t0
    shared_prev.drain()
    torch.cuda.synchronize(dev)

    if args.workload == "scale":
        # kernel duration from a second, event-bracketed pass: per-launch events would dominate the
        # wall clock of a ~10 us kernel, so they stay out of the pass `value` is computed from.
        ctx.profile_enable(True)
        for k in range(steps):
            step(k)
        torch.cuda.synchronize(dev)
    motion_stats = None
    if args.workload != "scale" and os.environ.get("LFG_MOTION_MODE", "0") != "1":
        motion_stats = ctx.motion_last_stats()        # after the timed region: it synchronises and copies counters
    stage_ms = {}
    for name, sid in (("scale", capi.STAGE_SCALE), ("motion", capi.STAGE_MOTION), ("interpolate", capi.STAGE_INTERPOLATE)):
        ms, n = ctx.profile_get(sid)
        if n:
            stage_ms[name] = ms / n
    ctx.profile_enable(False)

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    units_per_step = 1 if args.workload == "scale" else len(factors)
    in_res = args.workload == "pipeline_input_res"
    mw, mh = (w_in, h_in) if in_res else (w, h)        # resolution motion and interpolate run at
    value = world * steps * units_per_step / elapsed

    if rank == 0:
        stages = {}
        for name, avg in stage_ms.items():
            b = algorithmic_bytes(name, w_in, h_in, w if name == "scale" else mw, h if name == "scale" else mh)
            gbs = b / (avg * 1e-3) / 1e9
            stages[name] = {"avg_ms": round(avg, 5), "algorithmic_bytes": b, "hbm_gbs": round(gbs, 1),
                            "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
        dominant = max(stage_ms, key=stage_ms.get)
        if dominant == "motion":
            fl = motion_flops(mw, mh)
            tf = fl / (stage_ms["motion"] * 1e-3) / 1e12
            exact_only = os.environ.get("LFG_MOTION_MODE", "0") == "1"
            roofline = {"kernel": ("motion_tiled_8_16_kernel" if exact_only else
                                   "motion_prefilter_kernel (+ motion_hint/order_kernel, motion_resolve_kernel, motion_tiled_8_16_kernel on flagged tiles)"),
                        "bound": "valu", "achieved": round(tf, 2),
                        "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_VALU_PEAK_TFLOPS, 4),
                        "traffic": None,
                        "algorithmic_flops": fl,
                        "note": ("motion.comp is fp32-VALU bound, not HBM bound (SURVEY.md F7).  achieved = ALGORITHMIC flops of "
                                 "the shader (W*H*1089*(64 adds + 12 per distance)) / duration of the whole lfg_motion call (HIP "
                                 "events around its launches).  The default path does not execute all of them: an exact bracket "
                                 "on the cost rules out all but ~1 candidate per pixel before the literal 64-add chain is "
                                 "needed, and a partial-distortion test drops most candidates of a 16 x 56 pixel segment after "
                                 "24 of their 1449 distances (DESIGN.md, motion) -- which is how frac can exceed 1: it is the "
                                 "rate at which the shader's work is disposed of, not a utilisation (see `executed`).  "
                                 "LFG_MOTION_MODE=1 runs the literal kernel alone.  The 157.3 TFLOP/s peak counts an FMA as 2; "
                                 "an add-only stream tops out at 78.65.")}
            if not exact_only and args.input == "1080p" and not in_res:
                ex = pmc_executed("lfg::motion_prefilter")
                if ex is not None:
                    roofline["executed"] = ex
            if motion_stats is not None:
                roofline["motion_mode"] = "prefiltered"
                roofline["fallback_tiles"] = {"of": motion_stats[0], "exact_kernel": motion_stats[1]}
                roofline["candidates_recorded_per_pixel"] = round(motion_stats[2], 2)
            else:
                roofline["motion_mode"] = "exact kernel only"
        else:
            s = stages[dominant]
            roofline = {"kernel": {"scale": "scale_2x_kernel", "interpolate": "interpolate_kernel"}[dominant],
                        "bound": "hbm", "achieved": s["hbm_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": s["hbm_frac"], "traffic": None}
        size_name = {"540p": "540p->1080p", "1080p": "1080p->4K", "4k": "4K->8K", "8k": "8K->16K"}[args.input]
        if args.input == "1080p" and not in_res:
            t, src = pmc_traffic(["lfg::motion_tiled", "lfg::motion_prefilter", "lfg::motion_resolve", "lfg::motion_hint", "lfg::motion_order"] if dominant == "motion"
                                 else {"scale_2x_kernel": "lfg::scale_2x", "interpolate_kernel": "lfg::interpolate"}[roofline["kernel"]])
            if t is not None:
                roofline["traffic"] = t
                roofline["traffic_source"] = (f"{src}: FETCH_SIZE + WRITE_SIZE from two separate rocprofv3 --pmc passes, bytes per "
                                              "launch; FETCH_SIZE raw (uncalibrated for 4-byte-per-lane loads)")
        def launches(n):        # per step
            if n == "interpolate":
                return len(factors)
            if n == "scale":            # curr (+ the generated frames of the input-resolution variant) (+ the shared previous frame)
                return 1 + (len(factors) if in_res else 0) + (1 if share_input and args.workload == "pipeline" else 0)
            return 1
        total_bytes = sum(algorithmic_bytes(n, w_in, h_in, w if n == "scale" else mw, h if n == "scale" else mh) * launches(n)
                          for n in stage_ms)
        line = {
            "metric": (f"interpolated frames/s, {size_name} RGBA8" if args.workload == "pipeline"
                       else f"interpolated frames/s, {size_name} RGBA8 (variant: motion + interpolate at input resolution)" if in_res
                       else f"upscaled frames/s, {size_name} RGBA8 (Lanczos only)"),
            "value": round(value, 3),
            "unit": "frames/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 arithmetic on u8 RGBA (u8 in, u8 out; int8 motion vectors)",
            "data": "synthetic",
            "config": {"workload": (f"{args.input}->{2 * h_in}p " + ("scale+motion(8,16)+interpolate" if args.workload == "pipeline"
                                                                    else "motion(8,16)+interpolate at input resolution, then scale real and generated frame" if in_res
                                                                    else "scale only")),
                       "input": [w_in, h_in], "output": [w, h], "factors": factors if args.workload != "scale" else [],
                       "content": args.content, "parallelism": f"one frame pair per GPU x{world}" + (", RCCL broadcast of the shared previous input frame per step, upscaled on every rank" if world > 1 else "")},
            "roofline": roofline,
            "stages": stages,
            "path_hbm": {"algorithmic_bytes_per_step": total_bytes,
                         "achieved_gbs": round(total_bytes * steps / elapsed / 1e9 * 1.0, 1),
                         "frac_of_8TBs": round(total_bytes * steps / elapsed / 1e9 / HBM_PEAK_GBS, 5)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(w_in, h_in, w, h, factors, args.workload)
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
