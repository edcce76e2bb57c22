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

Frames in flight (--in-flight N, default 3; include/linuxfg_hip.h "Lanes"): step k runs on lane k % N of
the context -- its own stream, motion workspace and curr / mv / out buffers -- so that one step's upscale, hints and first
motion units fill the CUs that the previous step's last long motion units leave idle.  The dependencies of a stream of
frames are kept (a step's motion waits for the previous step's upscale on the other lane); `value` is still steps over
wall time between two device-wide synchronisations.  Per-stage durations (`stages`, `roofline.dominant_stage`) then come
from a second pass on ONE lane, where an event pair around a stage times that stage alone.  --in-flight 1 is strictly
one step at a time.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), launched by torch.distributed.run.
Frame pairs are independent, so the work shards one pair per GPU with no data-path collective except
the one the path really has: the batch's shared previous frame is broadcast from rank 0 each step
(double-buffered, issued one step ahead so it overlaps the kernels) -- as the 8.3 MB input frame, which
every rank upscales itself (one more 13 us scale per step than at N = 1; 33 MB of upscaled frame per
0.5 ms step would cost more on xGMI than that).  Weak scaling.

Besides the headline line's `value`, rank 0 at N = 1 measures in the same run (short, after the timed region): the
scale-only (BASELINE config 2) and scale + interpolate rates -- the two configurations SURVEY.md 8(d) says HBM is
the right bound for -- and `content_sweep`, the pipeline on every other synthetic content, because the motion
stage's run time depends on the content by more than an order of magnitude (results never do) and the default
content, a pure pan, is its best case.

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

# HIP gives a process four hardware queues by default and streams beyond that share them (two lanes on one queue run in turn:
# NOTES_r05.md section 7).  Read by the HIP runtime at its first call -- before torch or the library touch the GPU; an explicit setting wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md:36 (spec; 6290 measured copy)
FP32_VALU_PEAK_TFLOPS = 157.3  # ibid. :41 (counts an FMA as 2; an add-only stream tops out at 78.65)

SIZES = {"540p": (960, 540), "1080p": (1920, 1080), "4k": (3840, 2160), "8k": (7680, 4320)}


CONTENTS = ["translated", "occluded", "objects", "noisy", "uncorrelated", "static", "fade"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["pipeline", "scale", "pipeline_input_res"], default="pipeline")
    ap.add_argument("--input", choices=list(SIZES), default="1080p", help="input size; output is 2x")
    ap.add_argument("--factors", default="0.5", help="comma-separated interpolation factors per pair")
    ap.add_argument("--content", choices=CONTENTS, default="translated",
                    help="translated (default, SURVEY.md 8(d): curr = prev shifted by (3,-2)), occluded (the same with patches of fresh noise), objects (the same with patches that move on their own), noisy (the same with +-2 levels of noise everywhere), uncorrelated (independent noise frames), "
                         "static (curr = prev), fade (flat grey frames one level apart: every candidate ties at a "
                         "non-zero cost, the prefilter's worst case -- the rim tiles fall back to the literal kernel)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="frames in flight on one GPU (lanes of the C-ABI): each has its own stream, motion workspace and output "
                         "buffers, so one frame's scale / hints / first prefilter units fill the CUs that the previous frame's last "
                         "long units leave idle; 1 = strictly one frame at a time")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fused-motion-interpolate", action="store_true",
                    help="pipeline workload, one factor: lfg_interpolate_frames in the north-star order (lfg_set_fused_motion_interpolate: the motion "
                         "kernels write the generated frame themselves, no interpolate dispatch, no motion-vector frame); a labelled variant, not the default")
    ap.add_argument("--semantics", choices=["reference", "intended"], default="reference",
                    help="reference (default): the shaders as written, the parity contract; intended: the opt-in lfg_set_semantics mode (vectors displace by "
                         "pixels, ties resolve to the shortest vector) -- a labelled variant, never the headline")
    ap.add_argument("--stream", action="store_true",
                    help="only the changing-stream measurement (extras.stream; see measure_stream): K = 12 distinct pairs whose content changes every three steps")
    ap.add_argument("--config5", action="store_true", help="only BASELINE config 5 on this GPU, short (extras.config5; see measure_config5)")
    ap.add_argument("--with-communicator", action="store_true",
                    help="only the N > 1 step with a one-rank communicator on this GPU (extras.with_communicator; see measure_with_communicator)")
    ap.add_argument("--no-extras", action="store_true", help="skip scale_only / scale_interpolate / content_sweep (profiling runs)")
    return ap.parse_args()


def make_content(name: str, w_in: int, h_in: int, rank: int, content_rank: int):
    """(prev_in, curr_in) of one synthetic content at input resolution.  The previous frame is shared by all ranks (seed
    of stream 0) except for `fade`; a rank's current frame is derived from it."""
    from linux_fg_amd import synth
    prev_in = synth.make_prev(w_in, h_in, synth.BASE_SEED)
    rank_shift = rank_motion(content_rank)
    if name == "translated":
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
    elif name == "uncorrelated":
        curr_in = synth.noise_bytes(w_in, h_in, (synth.BASE_SEED + 7919 * (rank + 1)) & 0xFFFFFFFF)
    elif name == "occluded":                       # the translated pair with 24 patches of fresh noise (2 % of the frame)
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
        fresh = synth.noise_bytes(w_in, h_in, (synth.BASE_SEED + 104729 * (rank + 1)) & 0xFFFFFFFF)
        rng = np.random.default_rng(20240 + rank)
        for _ in range(24):
            pw, ph = int(rng.integers(w_in // 60, w_in // 12)), int(rng.integers(h_in // 60, h_in // 12))
            x0, y0 = int(rng.integers(40, w_in - 40 - pw)), int(rng.integers(40, h_in - 40 - ph))
            curr_in[y0:y0 + ph, x0:x0 + pw] = fresh[y0:y0 + ph, x0:x0 + pw]
    elif name == "objects":                        # the translated pair with 24 patches that move on their own
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
        rng = np.random.default_rng(30240 + rank)
        for _ in range(24):
            pw, ph = int(rng.integers(w_in // 60, w_in // 12)), int(rng.integers(h_in // 60, h_in // 12))
            x0, y0 = int(rng.integers(40, w_in - 40 - pw)), int(rng.integers(40, h_in - 40 - ph))
            dx, dy = int(rng.integers(-7, 8)), int(rng.integers(-7, 8))
            curr_in[y0:y0 + ph, x0:x0 + pw] = prev_in[y0 - dy:y0 - dy + ph, x0 - dx:x0 - dx + pw]
    elif name == "noisy":                          # the translated pair plus sensor-like noise: +-2 levels per channel
        curr_in = synth.translate(prev_in, rank_shift, synth.BASE_SEED + rank)
        amp = int(os.environ.get("LFG_BENCH_NOISE_AMP", "2"))      # (experiments: other amplitudes; the benchmark's content is +-2)
        n = synth.noise_bytes(w_in, h_in, (synth.BASE_SEED + 15485863 * (rank + 1)) & 0xFFFFFFFF) % (2 * amp + 1)
        curr_in = np.clip(curr_in.astype(np.int16) + n.astype(np.int16) - amp, 0, 255).astype(np.uint8)
    elif name == "static":
        curr_in = prev_in.copy()
    elif name == "fade":
        prev_in = np.full((h_in, w_in, 4), 100, np.uint8)
        curr_in = np.full((h_in, w_in, 4), 101, np.uint8)
    else:
        raise ValueError(name)
    return prev_in, curr_in


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


def interpolate_bytes_moved(mv: np.ndarray, factors, intended: bool = False) -> int:
    """Bytes the interpolate stage has to move for THIS vector field (literal semantics): the vectors in
    (2 B/px) and one frame out per factor (4 B/px) always; 4 B of prev and 4 B of curr only for the pixels whose displaced
    sample lies inside [0,1]^2 for some factor (interpolate.comp:17-20 returns vec4(0) without a fetch otherwise) -- the same
    fp32 sums as the shader.  On the benchmark's pan no sample does and the stage moves 6 of its 14 algorithmic B/px; a
    fraction of the roofline has to be computed from THESE bytes."""
    h, w = mv.shape[:2]
    f = np.float32
    uvx = ((np.arange(w, dtype=np.float32) + f(0.5)) / f(w))[None, :]
    uvy = ((np.arange(h, dtype=np.float32) + f(0.5)) / f(h))[:, None]
    mx, my = mv[..., 0].astype(np.float32), mv[..., 1].astype(np.float32)
    if intended:
        mx, my = mx / f(w), my / f(h)
    need_p = np.zeros((h, w), bool)
    need_c = np.zeros((h, w), bool)
    for t in factors:
        for need, scale in ((need_p, f(-t)), (need_c, f(1.0) - f(t))):
            sx, sy = uvx + mx * scale, uvy + my * scale
            need |= ~((sx < 0) | (sy < 0) | (sx > 1) | (sy > 1))
    return int(2 * w * h + 4 * w * h * len(factors) + 4 * int(need_p.sum()) + 4 * int(need_c.sum()))


def library_sha16():
    """First 16 hex digits of the sha256 of the HIP library this process loaded: profiles record it, and a profile
    taken with another build is not quoted."""
    import hashlib
    from linux_fg_amd import capi
    with open(capi.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def read_profile(pattern):
    """The newest committed counter table profiles/<pattern> (written by tools/pmc_per_step.py over
    `tools/run_stage.py pipeline N` under rocprofv3 --kernel-trace --pmc ...; bench.py cannot run the profiler on
    itself).  Returns (table, source, None) or (None, source, reason): a table taken with another build of the library
    (`# lib_sha16`) is not quoted.  table = {"kernels": {name: {...}}, "per_step": {COUNTER: value}}."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    if not files:
        return None, None, f"no profiles/{pattern}"
    src = os.path.relpath(files[-1], ROOT)
    sha, table = None, {"kernels": {}, "per_step": {}}
    for line in open(files[-1]):
        t = line.split()
        if line.startswith("# lib_sha16"):
            sha = t[2]
        elif t and t[0] == "kernel":
            table["kernels"][t[1]] = {t[i]: float(t[i + 1]) for i in range(2, len(t) - 1, 2)}
        elif t and t[0] == "per_step":
            table["per_step"] = {t[i]: float(t[i + 1]) for i in range(1, len(t) - 1, 2)}
    if sha != library_sha16():
        return None, src, f"{src} was collected with library {sha}, this run loaded {library_sha16()}: not quoted"
    return table, src, None


def pmc_traffic(prefix="lfg::"):
    """HBM-side bytes per pipeline step of the kernels whose names start with `prefix`: FETCH_SIZE + WRITE_SIZE (KB per
    launch, two separate rocprofv3 passes) x launches per step.  (bytes, source) or (None, reason)."""
    table, src, why = read_profile("r*_hbm_traffic_pmc.txt")
    if table is None:
        return None, why
    total = 0.0
    for k, v in table["kernels"].items():
        if k.startswith(prefix):
            if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
                return None, f"{src}: {k} lacks one of the two passes"
            total += (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0 * v["launches_per_step"]
    return int(total), src


def valu_cost_weights():
    """{kernel name: mean issue cost of one of its VALU wave-instructions, in plain-op units} from the newest committed
    profiles/r*_valu_cost_weights.txt (tools/valu_cost_histogram.py: the kernel's STATIC opcode histogram x the guide's issue
    costs) taken from the loaded library; {} otherwise."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_cost_weights.txt")))
    if not files:
        return {}, None
    sha, w = None, {}
    for line in open(files[-1]):
        t = line.split()
        if line.startswith("# lib_sha16"):
            sha = t[2]
        elif t and t[0] == "kernel":
            w[t[1]] = float(t[t.index("mean_cost") + 1])
    if sha != library_sha16():
        return {}, f"{os.path.relpath(files[-1], ROOT)} was made from library {sha}, not the loaded one"
    return w, os.path.relpath(files[-1], ROOT)


def pmc_executed(prefix="lfg::motion_"):
    """What the kernels whose names start with `prefix` executed per pipeline step, from the newest committed SQ
    counter table: wave-level VALU / LDS / SALU instruction counts, and the share of the chip's VALU issue slots they
    filled while they ran -- under THREE stated conventions: a SIMD-32 issues a plain wave64 VALU instruction in
    2 cycles once two waves are resident (MI355X_MICROARCH.md:54, the rate 157.3 TFLOP/s corresponds to): un-weighted, and
    weighted by the kernel's static opcode mix (transcendentals and packed f32 twice a plain op: valu_cost_weights); and the
    4 cycles per instruction that ONE wave alone sustains (round 4's figure, kept for comparison).  None if unusable."""
    table, src, why = read_profile("r*_sq_counters.txt")
    if table is None:
        return None
    weights, wsrc = valu_cost_weights()
    valu = lds = salu = us = weighted = 0.0
    have_w = bool(weights)
    for k, v in table["kernels"].items():
        if k.startswith(prefix) and "SQ_INSTS_VALU" in v:
            n = v["launches_per_step"]
            valu += v["SQ_INSTS_VALU"] * n; lds += v.get("SQ_INSTS_LDS", 0.0) * n; salu += v.get("SQ_INSTS_SALU", 0.0) * n
            us += v["mean_us"] * n
            wk = next((w for name, w in weights.items() if name.startswith(k) or k.startswith(name)), None)
            have_w = have_w and wk is not None
            weighted += v["SQ_INSTS_VALU"] * n * (wk or 1.0)
    if us <= 0:
        return None
    cycles = us * 1e-6 * 2.4e9 * 1024
    out = {"source": src, "kernels": prefix + ("*" if prefix.endswith("_") else ""), "kernel_us_per_step_in_that_pass": round(us, 2),
           "valu_wave_instructions": valu, "lds_wave_instructions": lds, "salu_wave_instructions": salu,
           "valu_issue_utilisation_simd32_2_cycles": round(valu * 2.0 / cycles, 3),
           "valu_issue_utilisation_cost_weighted": round(weighted * 2.0 / cycles, 3) if have_w else None,
           "valu_issue_utilisation_one_wave_alone_4_cycles": round(valu * 4.0 / cycles, 3),
           "cost_weights": wsrc,
           "how": "VALU wave-instructions (SQ_INSTS_VALU) x cycles per instruction / (kernel time x 2.4 GHz x 1024 SIMDs).  simd32_2_cycles: a SIMD-32 issues "
                  "a plain wave64 VALU instruction in 2 cycles once two waves are resident -- the rate the 157.3 TFLOP/s fp32 peak corresponds to "
                  "(MI355X_MICROARCH.md:54,:473); cost_weighted: the same, each kernel's count x the mean issue cost of its STATIC opcode mix "
                  "(issue costs measured on this chip at four waves per SIMD, tools/bench_dpp.hip: transcendental 3.7, packed f32 2.1, v_dot4 2.1, DPP arithmetic 2.7 plain ops; tools/valu_cost_histogram.py -- a proxy: the executed mix is not counted; the clock under load is nearer 1.9 - 2.0 GHz than the 2.4 GHz all three figures assume, which would raise each by a fifth); "
                  "one_wave_alone_4_cycles: what one wave alone sustains, round 4's convention"}
    return out


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


def visible_devices() -> int:
    """HIP devices the library sees (lfg_device_count), asked in a CHILD process: the process that starts the ranks must
    never initialise the GPU itself."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); from linux_fg_amd import capi; print(capi.load().lfg_device_count())" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    if out.returncode != 0:
        raise SystemExit("bench.py: cannot count the HIP devices: " + out.stderr.strip()[-400:])
    return int(out.stdout.strip().splitlines()[-1])


def launch_command(n: int, argv, port: int):
    """`python bench.py --gpus N ...` started by hand (or by a driver that does not wrap it): the command that runs the same
    arguments as N ranks, one per GPU, on this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(args, argv) -> int:
    """--gpus N > 1 without a launcher around us: start N ranks as a fresh child (torch.distributed.run), pass its output
    through and return its exit code.  This process never touches the GPU and never exec()s."""
    import subprocess
    share_gpu = os.environ.get("LFG_BENCH_SHARE_GPU") == "1"          # (rehearsal: several ranks on one card, no RCCL)
    if not share_gpu:
        have = visible_devices()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} HIP devices, found {have} (lfg_device_count); "
                  "one rank per GPU, RCCL wants a device per rank", file=sys.stderr, flush=True)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(launch_command(args.gpus, argv, free_port()), env=env).returncode


def measure_config5(torch, capi, dev, dev_index, n_lanes, steps=96, contents=("noisy", "objects", "occluded")):
    """BASELINE config 5 on this GPU, short: 4K -> 8K, t = 1/4, 1/2, 3/4 (three generated 8K frames per pair: scale, motion ONCE,
    one pass of lfg_interpolate_multi), the benchmark's pan -- and, shorter still, the contents of `contents` (the pan is the
    motion stage's best case but `static`).  A context of its own; frames in flight as the headline run."""
    w_in, h_in = SIZES["4k"]
    w, h = 2 * w_in, 2 * h_in
    factors = [0.25, 0.5, 0.75]
    prev_in, curr_in = make_content("translated", w_in, h_in, 0, 0)
    ctx = capi.Context(dev_index)
    if n_lanes > 1:
        ctx.lanes(n_lanes)
    def frame(width, height, fmt=capi.FORMAT_RGBA8):
        t = torch.empty((height, width, 4 if fmt == capi.FORMAT_RGBA8 else 2), dtype=torch.uint8, device=dev)
        return t, capi.Context.wrap(t.data_ptr(), width, height, fmt)
    t_pin = torch.from_numpy(prev_in).to(dev); f_pin = capi.Context.wrap(t_pin.data_ptr(), w_in, h_in, capi.FORMAT_RGBA8)
    t_cin = torch.from_numpy(curr_in).to(dev); f_cin = capi.Context.wrap(t_cin.data_ptr(), w_in, h_in, capi.FORMAT_RGBA8)
    t_p8, f_p8 = frame(w, h)
    lanes = [(frame(w, h), frame(w, h, capi.FORMAT_MV_S8X2), [frame(w, h) for _ in factors]) for _ in range(n_lanes)]
    ctx.scale(f_pin, f_p8)
    ctx.sync()
    def step(k, n):
        j = k % n
        (tc, fc), (tm, fm), outs = lanes[j]
        if n > 1:
            ctx.lane_select(j)
            ctx.lane_wait((k - 1) % n)
        ctx.scale(f_cin, fc)
        if n > 1:
            ctx.lane_mark()
        ctx.motion(f_p8, fc, fm, 8, 16.0)
        ctx.interpolate_multi(f_p8, fc, fm, [f for _, f in outs], factors)
    def timed(n_steps, n):
        ctx.sync(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(n_steps):
            step(k, n)
        ctx.sync(); torch.cuda.synchronize(dev)
        return time.perf_counter() - t0
    timed(8 * n_lanes, n_lanes)                    # (a fresh process: clocks, page tables, the lanes' verdicts)
    t = timed(steps, n_lanes)
    if n_lanes > 1:
        ctx.lane_select(0)
    ctx.profile_reset(); ctx.profile_enable(True)
    timed(max(4, steps // 12), 1)
    per = {}
    for name, sid in (("scale", capi.STAGE_SCALE), ("motion", capi.STAGE_MOTION), ("interpolate", capi.STAGE_INTERPOLATE)):
        ms, n = ctx.profile_get(sid)
        per[name] = round(ms / max(n, 1), 4)
    ctx.profile_enable(False)
    st = ctx.motion_last_stats()
    ws = ctx.motion_workspace_size(w, h)
    rim, groups = ctx.motion_plan()
    algo = algorithmic_bytes("scale", w_in, h_in, w, h) + algorithmic_bytes("motion", w_in, h_in, w, h) + \
        algorithmic_bytes("interpolate", w_in, h_in, w, h) + 4 * w * h * (len(factors) - 1)
    out = {"workload": "BASELINE config 5 on one GPU: 4K->8K, scale + motion(8,16) once + interpolate at t = 0.25, 0.5, 0.75 (one pass)",
           "steps": steps, "frames_in_flight": n_lanes, "pairs_per_s": round(steps / t, 2),
           "interpolated_frames_per_s": round(steps * len(factors) / t, 1), "ms_per_pair": round(t / steps * 1e3, 4),
           "stage_ms_one_call_at_a_time": per, "motion_workspace_bytes_per_lane": ws, "motion_plan_rim_split": rim,
           "prefilter_workgroups": groups, "fallback_tiles": st[1], "algorithmic_bytes_per_pair": algo,
           "hbm_frac": round(algo * steps / t / 1e9 / HBM_PEAK_GBS, 5)}
    by_content = {}
    for name in contents:
        p_in, c_in = make_content(name, w_in, h_in, 0, 0)
        t_pin.copy_(torch.from_numpy(p_in)); t_cin.copy_(torch.from_numpy(c_in))
        if n_lanes > 1:
            ctx.lane_select(0)
        ctx.scale(f_pin, f_p8)
        ctx.sync()
        timed(2 * n_lanes, n_lanes)
        tt = timed(max(24, steps // 4), n_lanes)
        if n_lanes > 1:
            ctx.lane_select(0)
        ctx.profile_reset(); ctx.profile_enable(True)
        timed(8, 1)
        ms, cnt = ctx.profile_get(capi.STAGE_MOTION)
        ctx.profile_enable(False)
        by_content[name] = {"interpolated_frames_per_s": round(max(24, steps // 4) * len(factors) / tt, 1), "ms_per_pair": round(tt / max(24, steps // 4) * 1e3, 4),
                            "motion_ms_one_call_at_a_time": round(ms / max(cnt, 1), 4), "fallback_tiles": ctx.motion_last_stats()[1], "steps": max(24, steps // 4)}
    out["other_contents"] = by_content
    ctx.close()
    return out


def measure_with_communicator(torch, capi, sharding, dev, dev_index, n_lanes, contents=("translated", "uncorrelated"), steps=200, probe_us=170, use_comm=True):
    """The step every rank runs at N > 1, WITH a communicator on this GPU: a context of its own whose streams are the library's (a
    communicator takes 8 CUs away from them: include/linuxfg_hip.h, lfg_comm_reserved_cus), a communicator of one rank, the double-
    buffered broadcast of the shared previous INPUT frame issued a step ahead through lfg_broadcast_frame / lfg_comm_wait as on rank 0
    of a node -- and, because one rank's ncclBroadcast launches nothing, lfg_comm_probe in its place: 8 workgroups of RCCL's device
    kernel's footprint that stay 170 us (8.3 MB at 50 GB/s of one xGMI link).  `probe_ms_behind_its_step`: host clock from the end of
    the step the stand-in was ordered behind to its own end, sampled without frames in flight around it.  Run in a process of its own
    (`bench.py --with-communicator`): what a rank is."""
    w_in, h_in = SIZES["1080p"]
    w, h = 2 * w_in, 2 * h_in
    ctx = capi.Context(dev_index)
    if n_lanes > 1:
        ctx.lanes(n_lanes)
    try:
        if use_comm:
            ctx.comm_init(1, 0, capi.Context.comm_unique_id())
    except capi.LfgError as e:
        ctx.close()
        return {"skipped": str(e)}
    def frame(width, height, fmt=capi.FORMAT_RGBA8):
        t = torch.empty((height, width, 4 if fmt == capi.FORMAT_RGBA8 else 2), dtype=torch.uint8, device=dev)
        return t, capi.Context.wrap(t.data_ptr(), width, height, fmt)
    slots = [frame(w_in, h_in) for _ in range(2)]
    t_cin, f_cin = frame(w_in, h_in)
    lanes = [(frame(w, h), frame(w, h), frame(w, h, capi.FORMAT_MV_S8X2), frame(w, h)) for _ in range(n_lanes)]

    def sync_comm():
        if use_comm:
            ctx.comm_sync()

    class Transport(sharding.CapiTransport):
        def issue(self, slot):
            super().issue(slot)
            if probe_us > 0:
                self.ctx.comm_probe(8, probe_us, every_lane=not self.lane_only)

    out = {"reserved_cus": ctx.comm_reserved_cus(), "prefilter_workgroups_at_most": None, "frames_in_flight": n_lanes, "steps": steps, "by_content": {},
           "how": "own context, library streams (CU-masked: 248 of 256 CUs), one-rank communicator; step = wait(broadcast k) + issue(broadcast k + 1 = "
                  "lfg_broadcast_frame_lane + lfg_comm_probe(8 workgroups, 170 us)) + scale(shared) + scale(curr) + motion + interpolate; the last entry "
                  "is the pan with the always-safe lfg_broadcast_frame, which orders the broadcast behind everything every lane has been given"}
    for name, lane_only in [(c, True) for c in contents] + [(contents[0], False)]:
        p_in, c_in = make_content(name, w_in, h_in, 0, 0)
        for t, _ in slots:
            t.copy_(torch.from_numpy(p_in))
        t_cin.copy_(torch.from_numpy(c_in))
        torch.cuda.synchronize(dev)
        bc = (sharding.SharedFrameBroadcaster(2, Transport(ctx, [f for _, f in slots], src=0, behind_selected_lane_only=lane_only), world_size=2, is_source=True)
              if use_comm else sharding.SharedFrameBroadcaster(2, None, world_size=1, is_source=True))      # (diagnostic: the same loop without any communicator call)
        def step(k, n):
            j = k % n
            (_, fp4), (_, fc4), (_, fmv), (_, fout) = lanes[j]
            if n > 1:
                ctx.lane_select(j)
                ctx.lane_wait((k - 1) % n)
            ctx.scale(slots[bc.acquire(k)][1], fp4)
            ctx.scale(f_cin, fc4)
            if n > 1:
                ctx.lane_mark()
            ctx.motion(fp4, fc4, fmv, 8, 16.0)
            ctx.interpolate(fp4, fc4, fmv, fout, 0.5)
        def timed(first, count, n):
            ctx.sync(); sync_comm()
            t0 = time.perf_counter()
            for k in range(first, first + count):
                step(k, n)
            ctx.sync(); sync_comm()
            return time.perf_counter() - t0
        n_steps = steps if name == "translated" else max(12, steps // 8)
        timed(0, 4 * n_lanes, n_lanes)
        t = timed(4 * n_lanes, n_steps, n_lanes)
        # the stand-in against the step it is ordered behind, one at a time: lane 0 runs a step, the probe is issued, lane 1 runs the next
        behind = []
        k0 = 4 * n_lanes + n_steps
        if n_lanes > 1:
            for r in range(3):
                ctx.sync(); sync_comm()
                step(k0, n_lanes); k0 += 1               # (its acquire issues the next broadcast + probe, ordered behind this step's lane ... )
                lane_a = ctx.lane_current()
                step(k0, n_lanes); k0 += 1               # ( ... and the next lane's step takes the chip)
                lane_b = ctx.lane_current()
                ctx.lane_select(lane_a); ctx.lane_sync()
                ta = time.perf_counter()
                sync_comm()
                tb = time.perf_counter()
                ctx.lane_select(lane_b); ctx.lane_sync()
                tc = time.perf_counter()
                behind.append([round((tb - ta) * 1e3, 3), round((tc - tb) * 1e3, 3)])
        bc.drain()
        ctx.sync(); sync_comm()
        out["by_content"][name if lane_only else name + ", broadcast ordered behind EVERY lane (lfg_broadcast_frame)"] = {"frames_per_s": round(n_steps / t, 1), "ms_per_step": round(t / n_steps * 1e3, 5), "steps": n_steps,
                                   "probe_ms_behind_its_step_and_ms_ahead_of_the_next": behind}
    out["prefilter_workgroups_at_most"] = ctx.motion_plan()[1]
    if n_lanes > 1:
        ctx.lane_select(0)
    if use_comm:
        ctx.comm_destroy()
    ctx.close()
    return out


STREAM_SEGMENTS = (("translated", 0), ("objects", 1), ("noisy", 2), ("translated", 7))     # (content, rank: its translation -- rank_motion)


def measure_stream(torch, capi, dev, dev_index, n_lanes, pairs_per_segment=3, rotations=10):
    """A stream that CHANGES: K = 12 distinct frame pairs in rotation, three of each segment of
    STREAM_SEGMENTS -- the pan, moving objects on another pan, sensor noise on a third, a fourth pan -- so that the content a lane's
    call meets differs from what the lane's previous call found every third step, and the three launch decisions that go by
    that previous call (lean kernel and plan, persistent grid, second pass: lfg_capi.cpp, motion_run) can be wrong.  Same work per
    step as `value` (scale -> motion -> interpolate at 1080p -> 4K, n_lanes frames in flight).  Two host loops:
      paced    before a lane is reused the host waits for that lane's previous frame (lfg_lane_sync) -- what a caller with
               n frames in flight does; a guess is then at most n calls old;
      unpaced  the host enqueues as fast as it can, as the headline loop does; a guess can be a whole queue old.
    Returns frames/s of both, the calls whose verdict came back and how many had been launched on a wrong guess
    (lfg_motion_prediction_stats), each segment's content alone in the paced loop, the harmonic mean of those (weighted by steps)
    and `verified`: the LAST step's vectors against the literal kernel, whole frame."""
    w_in, h_in = SIZES["1080p"]
    w, h = 2 * w_in, 2 * h_in
    ctx = capi.Context(dev_index)
    if n_lanes > 1:
        ctx.lanes(n_lanes)
    def frame(width, height, fmt=capi.FORMAT_RGBA8):
        t = torch.empty((height, width, 4 if fmt == capi.FORMAT_RGBA8 else 2), dtype=torch.uint8, device=dev)
        return t, capi.Context.wrap(t.data_ptr(), width, height, fmt)
    from linux_fg_amd import synth
    pairs = []                                   # (segment, curr_in frame, prev4 frame) + the tensors that keep them alive
    for seg, (content, crank) in enumerate(STREAM_SEGMENTS):
        for j in range(pairs_per_segment):
            # distinct pairs: each its own previous frame (another seed), its current frame derived from it as make_content does
            seed = (synth.BASE_SEED + 1000003 * (seg * pairs_per_segment + j + 1)) & 0xFFFFFFFF
            prev_in = synth.make_prev(w_in, h_in, seed)
            # (make_content derives curr from the shared stream-0 previous frame; here against this pair's own: the same edits)
            curr_in = prev_in.copy()
            if content in ("translated", "objects", "noisy"):
                curr_in = synth.translate(prev_in, rank_motion(crank), seed + 17)
            if content == "objects":
                rng = np.random.default_rng(30240 + seg * 16 + j)
                for _ in range(24):
                    pw, ph = int(rng.integers(w_in // 60, w_in // 12)), int(rng.integers(h_in // 60, h_in // 12))
                    x0, y0 = int(rng.integers(40, w_in - 40 - pw)), int(rng.integers(40, h_in - 40 - ph))
                    dx, dy = int(rng.integers(-7, 8)), int(rng.integers(-7, 8))
                    curr_in[y0:y0 + ph, x0:x0 + pw] = prev_in[y0 - dy:y0 - dy + ph, x0 - dx:x0 - dx + pw]
            if content == "noisy":
                n = synth.noise_bytes(w_in, h_in, (seed + 15485863) & 0xFFFFFFFF) % 5
                curr_in = np.clip(curr_in.astype(np.int16) + n.astype(np.int16) - 2, 0, 255).astype(np.uint8)
            tp = torch.from_numpy(prev_in).to(dev); fp = capi.Context.wrap(tp.data_ptr(), w_in, h_in, capi.FORMAT_RGBA8)
            tc = torch.from_numpy(np.ascontiguousarray(curr_in)).to(dev); fc = capi.Context.wrap(tc.data_ptr(), w_in, h_in, capi.FORMAT_RGBA8)
            tp4, fp4 = frame(w, h)
            ctx.scale(fp, fp4)
            pairs.append((seg, fc, fp4, (tp, tc, tp4)))
    ctx.sync()
    lanes = [(frame(w, h), frame(w, h, capi.FORMAT_MV_S8X2), frame(w, h)) for _ in range(n_lanes)]

    def run(order, paced, n_steps):
        """n_steps steps over the pairs of `order` in rotation; seconds"""
        ctx.sync(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(n_steps):
            _, fc, fp4, _ = pairs[order[k % len(order)]]
            j = k % n_lanes
            (tc4, fc4), (tm, fm), (to, fo) = lanes[j]
            if n_lanes > 1:
                ctx.lane_select(j)
            if paced and k >= n_lanes:
                ctx.lane_sync()                                 # frame k - n is done: its buffers are free, its verdict is in
            if n_lanes > 1:
                ctx.lane_wait((k - 1) % n_lanes)
            ctx.scale(fc, fc4)
            if n_lanes > 1:
                ctx.lane_mark()
            ctx.motion(fp4, fc4, fm, 8, 16.0)
            ctx.interpolate(fp4, fc4, fm, fo, 0.5)
        ctx.sync(); torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    everything = list(range(len(pairs)))
    n_steps = rotations * len(pairs)
    out = {"pairs": len(pairs), "segments": [f"{c} (translation of rank {r}: {rank_motion(r)})" for c, r in STREAM_SEGMENTS],
           "pairs_per_segment": pairs_per_segment, "steps": n_steps, "frames_in_flight": n_lanes}
    for label, paced in (("paced", True), ("unpaced", False)):
        run(everything, paced, 2 * len(pairs))
        before = ctx.motion_prediction_stats()
        t = run(everything, paced, n_steps)
        after = ctx.motion_prediction_stats()
        d = [a - b for a, b in zip(after, before)]
        out[label] = {"frames_per_s": round(n_steps / t, 1), "ms_per_step": round(t / n_steps * 1e3, 5),
                      "calls_whose_verdict_came_back": d[0],
                      "of_which_launched_on_a_wrong_guess": {"lean_kernel_and_plan": d[1], "persistent_grid": d[2], "second_pass_small_grid_but_tiles_flagged": d[3]}}
    # the last step's vectors (unpaced run: step n_steps - 1) against the literal kernel
    last = n_steps - 1
    _, fc, fp4, _ = pairs[everything[last % len(everything)]]
    (tc4, fc4), (tm, fm), _ = lanes[last % n_lanes]
    got = tm.cpu().numpy().view(np.int8).copy()
    tchk, fchk = frame(w, h, capi.FORMAT_MV_S8X2)
    if n_lanes > 1:
        ctx.lane_select(0)
    ctx.set_motion_mode(capi.MOTION_EXACT_ONLY)
    ctx.motion(fp4, fc4, fchk, 8, 16.0)
    ctx.set_motion_mode(capi.MOTION_PREFILTERED)
    ctx.sync()
    diff = int((got != tchk.cpu().numpy().view(np.int8)).any(-1).sum())
    out["verified"] = {"step": last, "vectors_vs_literal_kernel": {"pixels": w * h, "differing": diff}, "ok": diff == 0}
    # every segment's content alone (its own three pairs in rotation), paced: what the stream would run at if nothing were ever guessed wrong
    alone, weights = {}, {}
    for seg, (content, crank) in enumerate(STREAM_SEGMENTS):
        mine = [i for i, p in enumerate(pairs) if p[0] == seg]
        run(mine, True, 2 * len(mine) * n_lanes)
        n = max(30, n_steps // 4)
        t = run(mine, True, n)
        alone[f"{seg}: {content}"] = round(n / t, 1)
        weights[f"{seg}: {content}"] = len(mine)
    hm = sum(weights.values()) / sum(weights[k] / alone[k] for k in alone)
    out["segments_alone_paced_frames_per_s"] = alone
    out["harmonic_mean_of_segments_alone"] = round(hm, 1)
    out["paced_over_harmonic_mean"] = round(out["paced"]["frames_per_s"] / hm, 4)
    out["how"] = ("K distinct pairs, content changing every %d steps; `paced` waits for a lane's previous frame before reusing the lane (lfg_lane_sync), `unpaced` "
                  "enqueues ahead like the headline loop; wrong guesses counted by the library for the calls whose verdict word it read back" % pairs_per_segment)
    ctx.close()
    return out



def measure_pcie_inclusive(torch, capi, ctx, dev, w_in, h_in, w, h):
    """The PCIe-INCLUSIVE second metric of SURVEY.md 8(f) rank 3 -- never `value`.  The reference uploads every captured frame
    through a staging buffer (src/window_capture.cpp:472-568) and reads every presented frame back (src/scaler.cpp:479-536);
    here: (a) what the host link gives a pinned 1080p upload and a pinned 4K read-back through the C-ABI, alone; (b) the C++
    host mirror's loop (linux-fg_amd/lfg_host: Scaler::ProcessFrame with the pinned ring, three frames in flight), which uploads
    one input frame and reads back TWO 4K frames (real + generated) per input frame -- with the host's frame synthesis in the
    loop as in round 2's figure (390 presented frames/s), and played back from memory with a presenter that looks at no pixel."""
    import subprocess
    out = {"host_link_spec_gbs": 63.0}
    n_in, n_out = w_in * h_in * 4, w * h * 4
    pin_in, pin_out = ctx.staging_create(n_in), ctx.staging_create(n_out)
    t_in = torch.empty((h_in, w_in, 4), dtype=torch.uint8, device=dev); f_in = capi.Context.wrap(t_in.data_ptr(), w_in, h_in, capi.FORMAT_RGBA8)
    t_out = torch.empty((h, w, 4), dtype=torch.uint8, device=dev); f_out = capi.Context.wrap(t_out.data_ptr(), w, h, capi.FORMAT_RGBA8)
    def rate(fn, nbytes, n=60):
        for _ in range(5):
            fn()
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(n):
            fn()
        ctx.sync()
        return round(nbytes * n / (time.perf_counter() - t0) / 1e9, 2)
    out["pinned_upload_1080p_gbs"] = rate(lambda: ctx.upload_async(f_in, pin_in), n_in)
    out["pinned_readback_4k_gbs"] = rate(lambda: ctx.download_async(f_out, pin_out), n_out)
    ctx.staging_destroy(pin_in); ctx.staging_destroy(pin_out)
    host = os.path.join(ROOT, "linux-fg_amd", "lfg_host")
    if not os.path.exists(host):
        out["host_loop"] = "linux-fg_amd/lfg_host not built"
        return out
    def run(extra, frames):
        cmd = [host, "--input-width", str(w_in), "--input-height", str(h_in), "--output-width", str(w), "--output-height", str(h),
               "--frames", str(frames), "--in-flight", "3", "--quiet"] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return {"error": (r.stderr or r.stdout).strip()[-300:]}
        d = json.loads(r.stdout.strip().splitlines()[-1])
        per_input = n_in + 2 * n_out                      # one upload, two read-backs per input frame
        d["readback_gbs"] = round(d["presented_fps"] * n_out / 1e9, 2)
        d["link_bytes_per_input_frame"] = per_input
        return {k: d[k] for k in ("input_frames", "presented", "interpolated", "seconds", "presented_fps", "readback_gbs", "link_bytes_per_input_frame", "note")}
    out["host_loop_with_frame_synthesis"] = run([], 120)
    out["host_loop_replayed"] = run(["--replay", "8", "--present-null"], 600)
    out["why"] = ("round 2's 390 presented frames/s (13 GB/s of read-back) was the HOST's synthetic frame source -- a hash and a 4-byte copy per pixel on one "
                  "core, 5 ms per 1080p frame -- plus a strided checksum over every presented frame, not the link: played back from memory with a presenter that "
                  "looks at nothing, the same loop is bounded by the two 33 MB read-backs per input frame (`pinned_readback_4k_gbs`)")
    out["how"] = "subprocess: lfg_host --in-flight 3 (Scaler::ProcessFrame, pinned ring, copy stream); a labelled second metric, never `value`"
    return out



def gather_ranks(dist, world, dev_index, rccl_ranks, verified):
    """What rank 0 needs from the other ranks for the line, over the control plane (gloo): which card every rank drove, the size of
    the RCCL communicator every rank joined (the driver checks its N against these), and every rank's check of its own frames."""
    if world <= 1 or dist is None:
        return [dev_index], rccl_ranks, verified
    gathered = [None] * world
    dist.all_gather_object(gathered, (dev_index, rccl_ranks, verified))
    devices = [g[0] for g in gathered]
    ranks = min(g[1] for g in gathered)                       # (every rank must report the same communicator size)
    if verified is not None:
        oks = [bool(g[2] is not None and g[2].get("ok")) for g in gathered]
        verified = dict(verified, ok=all(oks), per_rank_ok=oks)
    return devices, ranks, verified


def assemble_line(r):
    """The ONE JSON line rank 0 prints, from plain values (`r`: a namespace filled by main(); no GPU, no torch in here, so that
    tests/test_bench_helpers.py can build the line of an N > 1 run on CPU ranks and hold its keys)."""
    args, world, stage_ms, factors = r.args, r.world, r.stage_ms, r.factors
    w_in, h_in, w, h, mw, mh, in_res, share_input = r.w_in, r.h_in, r.w, r.h, r.mw, r.mh, r.in_res, r.share_input
    steps, warmup, elapsed, regions, value = r.steps, r.warmup, r.elapsed, r.regions, r.value
    exact_only, motion_stats, n_lanes, devices, rccl_ranks = r.exact_only, r.motion_stats, r.n_lanes, r.devices, r.rccl_ranks
    fused_mi, extras, stage_pass = r.fused_mi, r.extras, r.stage_pass
    stages = {}
    for name, avg in stage_ms.items():
        b = algorithmic_bytes(name, w_in, h_in, w if name == "scale" else mw, h if name == "scale" else mh)
        if name == "interpolate" and len(factors) > 1 and not in_res:
            b += 4 * mw * mh * (len(factors) - 1)               # one pass: inputs once, one frame per factor
        entry = {"avg_ms": round(avg, 5), "algorithmic_bytes": b}
        if name == "interpolate":
            # the stage moves its algorithmic bytes only where it samples (interpolate_bytes_moved): the fraction is of the bytes
            # it moved on THIS content's vectors (the pan: 6 of 14 B/px); `interpolate_alone.static` has the stage where it samples
            if r.interp_moved is not None:
                b = r.interp_moved
                entry["bytes_moved"] = b
        gbs = b / (avg * 1e-3) / 1e9
        entry.update({"hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)})
        stages[name] = entry
    dominant = max(stage_ms, key=stage_ms.get)

    def launches(n):        # per step
        if n == "interpolate":
            return len(factors) if in_res else 1
        if n == "scale":            # curr (+ the generated frames of the input-resolution variant) (+ the shared previous frame)
            return 1 + (len(factors) if in_res else 0) + (1 if share_input and args.workload == "pipeline" else 0)
        return 1
    total_bytes = sum(stages[n]["algorithmic_bytes"] * launches(n) for n in stage_ms)
    path_gbs = total_bytes * steps / elapsed / 1e9

    # `roofline`: the PATH against the HBM roofline -- algorithmic bytes of every stage of a step / ms_per_step.
    # It follows from this line alone and is a fraction.  The dominant kernel's own figures sit beside it.
    roofline = {"kernel": ("scale_2x_kernel" if args.workload == "scale" else
                           f"whole step: scale_2x_kernel -> lfg_motion's kernels -> interpolate kernel (dominant stage: {dominant})"),
                "bound": "hbm", "achieved": round(path_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(path_gbs / HBM_PEAK_GBS, 5), "traffic": None,
                "algorithmic_bytes_per_step": total_bytes,
                "how": "algorithmic bytes of a step (SURVEY.md 8(d): every stage reads each input once and writes each output once) "
                       "/ ms_per_step, against the 8 TB/s spec peak"}
    if args.workload != "scale" and "motion" in stage_ms:
        fl = motion_flops(mw, mh)
        md = {"kernel": ("motion_tiled_8_16_kernel" if exact_only else "the kernels of one lfg_motion call"),
              "avg_ms": stages["motion"]["avg_ms"], "hbm_frac": stages["motion"]["hbm_frac"],
              "work_disposed_tflops": round(fl / (stage_ms["motion"] * 1e-3) / 1e12, 2), "algorithmic_flops": fl,
              "note": ("motion.comp is fp32-VALU bound, not HBM bound (SURVEY.md F7): 1089 candidates x 64 block positions per "
                       "pixel.  work_disposed_tflops = the shader's algorithmic flops (W*H*1089*(64 adds + 12 per distance)) / "
                       "the duration of the lfg_motion call: the rate at which that work is DISPOSED OF, not executed -- an exact "
                       "bracket on the cost rules out all but ~1 candidate per pixel and a partial-distortion test drops most "
                       "candidates of a segment after 14 of their 1449 distances (DESIGN.md, motion) -- so it may exceed the "
                       "157.3 TFLOP/s fp32 peak and is NOT a utilisation; `executed` (when a counter profile of this very "
                       "library is committed) is.  LFG_MOTION_MODE=1 runs the literal kernel alone.")}
        if not exact_only and args.input == "1080p" and not in_res:
            ex = pmc_executed("lfg::motion_prefilter_kernel")           # the dominant KERNEL of the dominant stage
            if ex is not None:
                md["executed"] = ex
                ex_all = pmc_executed("lfg::motion_")
                if ex_all is not None:
                    md["executed_all_motion_kernels"] = {k: ex_all[k] for k in ("kernels", "kernel_us_per_step_in_that_pass", "valu_issue_utilisation_simd32_2_cycles",
                                                                                "valu_issue_utilisation_cost_weighted", "valu_issue_utilisation_one_wave_alone_4_cycles")}
        if motion_stats is not None:
            md["motion_mode"] = "prefiltered"
            md["fallback_tiles"] = {"of": motion_stats[0], "exact_kernel": motion_stats[1]}
            md["candidates_recorded_per_pixel"] = round(motion_stats[2], 2)
        else:
            md["motion_mode"] = "exact kernel only"
        roofline["dominant_stage"] = md
    size_name = {"540p": "540p->1080p", "1080p": "1080p->4K", "4k": "4K->8K", "8k": "8K->16K"}[args.input]
    if args.input == "1080p" and not in_res and len(factors) == 1:
        t, src = pmc_traffic("lfg::scale_2x" if args.workload == "scale" else "lfg::")
        if t is not None:
            roofline["traffic"] = t
            roofline["traffic_source"] = (f"{src}: FETCH_SIZE + WRITE_SIZE of every kernel of a step, two separate rocprofv3 "
                                          "--pmc passes with this very library; FETCH_SIZE raw (uncalibrated for 4-byte-per-lane loads)")
        else:
            roofline["traffic_note"] = src
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
                   "order": ("north-star, fused: lfg_interpolate_frames with lfg_set_fused_motion_interpolate -- the motion kernels write the generated frame, "
                             "no interpolate dispatch, no motion-vector frame (a labelled variant)" if fused_mi else "one call per stage"),
                   "semantics": args.semantics,       # "reference": the shaders as written; "intended": the opt-in lfg_set_semantics mode (a labelled variant)
                   "frames_in_flight": n_lanes,       # lanes of the C-ABI; 1 = strictly one frame at a time
                   "devices": devices,                # HIP device ordinal of every rank, in rank order
                   "content": args.content + (" (the motion stage's best case but `static`; see content_sweep)" if args.content == "translated" and args.workload != "scale" else ""),
                   "parallelism": f"one frame pair per GPU x{world}" + (", lfg_broadcast_frame_lane (RCCL, 8 CUs kept for it) of the shared previous input frame per step, upscaled on every rank" if world > 1 else ""),
                   "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},      # (HIP's default of 4 makes streams share hardware queues)
        "rccl_ranks": rccl_ranks,              # lfg_comm_ranks(): the size of the RCCL communicator every rank joined (0: one GPU, none)
        "repeats": {"regions": len(regions), "steps_per_region": steps, "median_ms_per_step": round(elapsed / steps * 1e3, 5),
                    "min_ms_per_step": round(min(regions) / steps * 1e3, 5), "max_ms_per_step": round(max(regions) / steps * 1e3, 5),
                    "first_region_ms_per_step": round(regions[0] / steps * 1e3, 5),
                    "how": "every region is exactly `steps` steps between its own barrier + device-synchronisation pairs (max over ranks); "
                           "a region shorter than 0.25 s is repeated until 0.25 s have been timed (25 regions at most) and `value` / "
                           "`ms_per_step` are the MEDIAN region's"},
        "roofline": roofline,
        "stages": stages,
        "stages_how": stage_pass,
        "library_sha16": library_sha16(),
    }
    line["verified"] = r.verified          # what the timed kernels produced, checked after the timed regions (every rank's; None: not applicable)
    line.update(extras)
    # the CPU baseline is timed on rank 0 at N = 1 only (the driver's contract); the key is there at every N
    line["cpu_baseline"] = r.cpu_baseline() if (world == 1 and r.cpu_baseline is not None) else None
    return line



def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1:                 # --gpus given and contradicted by the launcher: refuse, rather than print another N's figure
            raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they have to agree")
        args.gpus = world                  # (launched by torch.distributed.run without --gpus: its world size is the number of GPUs)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # One rank per GPU.  torch.distributed (gloo, CPU tensors) is the CONTROL plane only: it carries the 128-byte
    # communicator id, the barriers and the max-over-ranks of the elapsed time.  The data path's one collective --
    # the shared previous frame -- is the C-ABI's lfg_broadcast_frame (RCCL over xGMI).  LFG_BENCH_SHARE_GPU=1
    # (rehearsal only) lets several ranks share one card with torch carrying the frame too: RCCL itself needs one
    # device per rank.
    share_gpu = os.environ.get("LFG_BENCH_SHARE_GPU") == "1"
    from linux_fg_amd import capi as _capi
    n_dev = _capi.load().lfg_device_count()
    if world > 1 and not share_gpu and n_dev < world:
        raise SystemExit(f"bench.py: --gpus {world} needs {world} HIP devices, found {n_dev} (lfg_device_count): one rank per GPU")
    dev_index = local_rank % torch.cuda.device_count() if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from linux_fg_amd import capi, sharding, synth

    if args.stream:                        # the changing-stream measurement alone (the default run carries it in `stream`)
        print(json.dumps({"stream": measure_stream(torch, capi, dev, dev_index, max(1, min(args.in_flight, capi.MAX_LANES))),
                          "library_sha16": library_sha16()}), flush=True)
        return

    if args.config5:
        print(json.dumps({"config5": measure_config5(torch, capi, dev, dev_index, max(1, min(args.in_flight, capi.MAX_LANES))),
                          "library_sha16": library_sha16()}), flush=True)
        return
    if args.with_communicator:             # ... and the N > 1 step with a communicator alone (the default run starts this as a process of its own)
        print(json.dumps({"with_communicator": measure_with_communicator(torch, capi, sharding, dev, dev_index, max(1, min(args.in_flight, capi.MAX_LANES))),
                          "library_sha16": library_sha16()}), flush=True)
        return

    w_in, h_in = SIZES[args.input]
    w, h = 2 * w_in, 2 * h_in
    factors = [float(x) for x in args.factors.split(",") if x]
    # defaults: a timed region of about a second (0.46 ms per pipeline step, 12 us per scale step at 1080p -> 4K)
    steps = args.steps if args.steps is not None else (40000 if args.workload == "scale" else 2000)
    warmup = args.warmup if args.warmup is not None else (500 if args.workload == "scale" else 20)

    ctx = capi.Context(dev_index)
    stream = torch.cuda.current_stream(dev)
    if world > 1 and not share_gpu:
        # With a communicator the library's OWN streams carry the CU mask that keeps 8 CUs for RCCL's kernel (include/linuxfg_hip.h:
        # lfg_comm_reserved_cus); torch's current stream has none, and masked streams synchronise with the NULL stream: lane 0 stays on
        # the context's own stream.  (Every hand-over between torch and the library below is a device-wide synchronise.)
        ctx.comm_init(world, rank, sharding.exchange_comm_id(dist, capi.Context.comm_unique_id, src=0))
    else:
        ctx.set_stream(stream.cuda_stream)

    def dev_frame(host: np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(host)).to(dev)
        return t, capi.Context.wrap(t.data_ptr(), host.shape[1], host.shape[0], capi.FORMAT_RGBA8)

    def empty_frame(width, height, fmt=capi.FORMAT_RGBA8):
        ch = 4 if fmt == capi.FORMAT_RGBA8 else 2
        t = torch.empty((height, width, ch), dtype=torch.uint8, device=dev)
        return t, capi.Context.wrap(t.data_ptr(), width, height, fmt)

    # Synthetic inputs, uploaded once before timing.  The previous frame is shared by all ranks (seed of stream 0);
    # each rank's current frame is that frame translated by its own vector, inside the search range after the 2x
    # upscale (|2 dx|, |2 dy| <= 16): (3, -2) on rank 0, then (4, -2) ... (7, -2), (3, -3) ...
    content_rank = int(os.environ.get("LFG_BENCH_CONTENT_RANK", rank))       # (diagnostic: another rank's frames on this GPU)
    prev_in, curr_in = make_content(args.content, w_in, h_in, rank, content_rank)
    t_prev_in, f_prev_in = dev_frame(prev_in)
    t_curr_in, f_curr_in = dev_frame(curr_in)
    t_curr4, f_curr4 = empty_frame(w, h)
    outs = [empty_frame(w, h) for _ in factors]
    f_outs = [f for _, f in outs]
    t_mv, f_mv = empty_frame(w, h, capi.FORMAT_MV_S8X2)
    if args.workload == "pipeline_input_res":
        t_mv_in, f_mv_in = empty_frame(w_in, h_in, capi.FORMAT_MV_S8X2)
    # The previous frame.  One GPU: upscaled once, before the timed region (in a stream it is the last step's current
    # frame).  Several GPUs: the batch shares its previous frame, which travels as the 8.3 MB INPUT frame -- one
    # broadcast per step, double-buffered -- and every rank upscales it (13 us) rather than 33 MB of upscaled frame
    # crossing xGMI under a 0.8 ms step.  LFG_BENCH_SHARE_INPUT=1 runs that data flow on one GPU (no collective).
    share_input = world > 1 or os.environ.get("LFG_BENCH_SHARE_INPUT") == "1"
    t_prev4, f_prev4 = empty_frame(w, h)
    ctx.scale(f_prev_in, f_prev4)
    prev_slots = [dev_frame(prev_in) for _ in range(2)] if share_input else [(t_prev4, f_prev4)]
    torch.cuda.synchronize(dev)

    transport = None
    if world > 1:
        # (behind the selected lane only: the step has waited for the previous step's upscales -- the last readers of the slot -- by then)
        transport = (sharding.TorchTransport(dist, [t for t, _ in prev_slots], src=0) if share_gpu
                     else sharding.CapiTransport(ctx, [f for _, f in prev_slots], src=0, behind_selected_lane_only=True))
    shared_prev = sharding.SharedFrameBroadcaster(len(prev_slots), transport, world_size=world, is_source=rank == 0)

    # Frames in flight (`pipeline` workload; include/linuxfg_hip.h "Lanes").  Step k runs on lane k % n: its own stream,
    # motion workspace and prev4 / curr4 / mv / out buffers.  As in a stream of frames, a step waits for the previous
    # step's upscales -- on the other lane -- and for nothing else of it; with several GPUs that wait also frees the
    # broadcast slot the next shared frame is about to land in.
    n_lanes = max(1, min(args.in_flight, capi.MAX_LANES)) if args.workload == "pipeline" else 1
    lane_bufs = [(f_curr4, f_mv, f_outs, f_prev4)]
    lane_tensors = [(t_mv, [t for t, _ in outs])]      # (the same buffers as torch tensors: `verified` reads them back)
    keep_alive = []
    if n_lanes > 1:
        ctx.lanes(n_lanes)
        for j in range(1, n_lanes):
            bufs = [empty_frame(w, h), empty_frame(w, h, capi.FORMAT_MV_S8X2)] + [empty_frame(w, h) for _ in factors]
            fp4 = f_prev4
            if share_input:                              # the shared previous frame is upscaled per step: a buffer per lane
                bufs.append(empty_frame(w, h))
                fp4 = bufs[-1][1]
            keep_alive.append(bufs)
            lane_bufs.append((bufs[0][1], bufs[1][1], [f for _, f in bufs[2:2 + len(factors)]], fp4))
            lane_tensors.append((bufs[1][0], [t for t, _ in bufs[2:2 + len(factors)]]))

    fused_mi = bool(args.fused_motion_interpolate) and args.workload == "pipeline" and len(factors) == 1
    if args.semantics == "intended":
        ctx.set_semantics(capi.SEMANTICS_INTENDED)
    if fused_mi:
        ctx.set_fused_motion_interpolate(True)

    def interpolate_all(fp, fc, fm, fouts=None):
        fouts = f_outs if fouts is None else fouts
        if len(factors) == 1:
            ctx.interpolate(fp, fc, fm, fouts[0], factors[0])
        else:
            ctx.interpolate_multi(fp, fc, fm, fouts, factors)        # one pass over prev / curr / mv, every factor

    def pipeline_step(k, fc_in, lanes, shared=None, bufs=None, slots=None):
        """scale -> motion -> interpolate of step k on lane k % lanes (shared: the step's shared previous INPUT frame)"""
        j = k % lanes
        fc4, fmv, fouts, fp4 = (lane_bufs if bufs is None else bufs)[j]
        if lanes > 1:
            ctx.lane_select(j)
            ctx.lane_wait((k - 1) % lanes)
        if shared is not None:
            f_shared = (prev_slots if slots is None else slots)[shared.acquire(k)][1]             # (waits for its broadcast, issues the next one)
            ctx.scale(f_shared, fp4)
        ctx.scale(fc_in, fc4)
        if lanes > 1:
            ctx.lane_mark()
        if fused_mi:
            ctx.interpolate_frames(fp4, fc4, fouts[0], factors[0])   # motion + interpolate in one call: the motion kernels write the frame
            return
        ctx.motion(fp4, fc4, fmv, 8, 16.0)
        interpolate_all(fp4, fc4, fmv, fouts)

    def step(k):
        if args.workload == "pipeline":
            pipeline_step(k, f_curr_in, n_lanes, shared_prev if share_input else None)
            return
        # the shared previous frame of this step (waits for its broadcast, issues the next one)
        f_shared = prev_slots[shared_prev.acquire(k)][1]
        f_prev_step = f_shared if share_input else f_prev_in
        if args.workload == "pipeline_input_res":
            ctx.motion(f_prev_step, f_curr_in, f_mv_in, 8, 16.0)
            ctx.scale(f_curr_in, f_curr4)
            for t, fo in zip(factors, f_outs):      # one kernel: interpolated row by row inside the 2x scale kernel
                ctx.interpolate_scale(f_prev_step, f_curr_in, f_mv_in, fo, t)
            return
        ctx.scale(f_curr_in, f_curr4)

    def barrier_sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(fn, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(n):
            fn(k)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0

    shared_prev.start(0)
    for k in range(warmup):
        step(k)
    ctx.profile_reset()
    ctx.profile_enable(args.workload != "scale" and n_lanes == 1)     # events around every stage launch (ms-scale kernels)

    def timed_region(first):
        """EXACTLY `steps` steps between two barrier + device-synchronisation pairs; seconds, max over ranks."""
        barrier_sync()
        t0 = time.perf_counter()
        for k in range(first, first + steps):
            step(k)
        barrier_sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # A short region (the driver asks for 20 steps: 7 ms) is mostly ramp-up of the frames in flight and timer noise, so it is
    # REPEATED -- every region is exactly `steps` steps between its own barriers -- until a quarter of a second has been
    # timed (25 regions at most) and the MEDIAN region gives `value`; `repeats` reports the spread and the first region.
    regions = [timed_region(warmup)]
    n_regions = 1 if regions[0] >= 0.25 else int(min(25, max(3, np.ceil(0.25 / max(regions[0], 1e-6)))))
    for r in range(1, n_regions):
        regions.append(timed_region(warmup + r * steps))
    elapsed = float(np.median(regions))
    shared_prev.drain()
    torch.cuda.synchronize(dev)

    # ---- what the timed kernels produced, looked at (no oracle: the HIP path against itself and against what the content
    # implies).  The LAST timed step's vectors and generated frame, as they lie in its lane's buffers, must equal -- byte for
    # byte, the whole frame -- an un-timed run of the same step with the literal motion kernel alone (lfg_set_motion_mode:
    # every (pixel, candidate) through the shader's own 64-term chain); and on the benchmark's pan the vectors away from the
    # rim must be minus the translation.
    verified = None
    if args.workload == "pipeline" and not fused_mi and os.environ.get("LFG_MOTION_MODE", "0") != "1":
        last = warmup + len(regions) * steps - 1
        t_mv_last, t_outs_last = lane_tensors[last % n_lanes]
        got_mv = t_mv_last.cpu().numpy().view(np.int8).copy()
        got_out = [t.cpu().numpy().copy() for t in t_outs_last]
        chk_mv_t, chk_mv = empty_frame(w, h, capi.FORMAT_MV_S8X2)
        chk_outs = [empty_frame(w, h) for _ in factors]
        fc4_last, _, _, fp4_last = lane_bufs[last % n_lanes]
        if n_lanes > 1:
            ctx.lane_select(0)
        ctx.set_motion_mode(capi.MOTION_EXACT_ONLY)
        ctx.motion(fp4_last, fc4_last, chk_mv, 8, 16.0)
        interpolate_all(fp4_last, fc4_last, chk_mv, [f for _, f in chk_outs])
        ctx.set_motion_mode(capi.MOTION_PREFILTERED)
        torch.cuda.synchronize(dev)
        want_mv = chk_mv_t.cpu().numpy().view(np.int8)
        mv_diff = int((got_mv != want_mv).any(-1).sum())
        out_diff = int(sum((g != t.cpu().numpy()).any(-1).sum() for g, (t, _) in zip(got_out, chk_outs)))
        verified = {"step": last, "lane": last % n_lanes,
                    "vectors_vs_literal_kernel": {"pixels": w * h, "differing": mv_diff},
                    "generated_frames_vs_literal_kernel": {"pixels": w * h * len(factors), "differing": out_diff},
                    "how": "the last timed step's motion vectors and generated frame(s), read back from its lane's buffers, against an un-timed "
                           "run of the same step through the literal motion kernel alone (LFG_MOTION_EXACT_ONLY), whole frames, byte for byte"}
        ok = mv_diff == 0 and out_diff == 0
        if args.content == "translated":
            sx, sy = rank_motion(content_rank)
            inner = got_mv[64:h - 64, 64:w - 64]
            off = int(((inner[..., 0] != -2 * sx) | (inner[..., 1] != -2 * sy)).sum())
            verified["translation_property"] = {"expected_vector": [-2 * sx, -2 * sy], "interior_pixels": int(inner.shape[0] * inner.shape[1]), "differing": off}
            ok = ok and off == 0
        if args.content == "static":
            still = int((got_mv != 0).any(-1).sum())
            verified["static_property"] = {"expected_vector": [0, 0], "differing": still}
            ok = ok and still == 0
        verified["ok"] = bool(ok)
        del chk_mv_t, chk_outs

    if n_lanes > 1:
        # stage durations from a second pass on ONE lane, event-bracketed: with frames in flight the stages of two steps
        # overlap and an event pair around one of them times both
        ctx.lane_select(0)
        ctx.profile_enable(True)
        for k in range(min(steps, 200)):
            pipeline_step(k, f_curr_in, 1)           # (the shared previous frame, if any, stays as the last step left it)
        torch.cuda.synchronize(dev)
    if args.workload == "scale":
        # kernel duration from a second, event-bracketed pass: per-launch events would dominate the
        # wall clock of a ~10 us kernel, so they stay out of the pass `value` is computed from.
        ctx.profile_enable(True)
        for k in range(min(steps, 2000)):
            step(k)
        torch.cuda.synchronize(dev)
    motion_stats = None
    exact_only = os.environ.get("LFG_MOTION_MODE", "0") == "1"
    if args.workload != "scale" and not exact_only:
        motion_stats = ctx.motion_last_stats()        # after the timed region: it synchronises and copies counters
    rim_split, prefilter_groups = ctx.motion_plan()
    stage_pass = {"plan_rim_split": rim_split, "prefilter_workgroups": prefilter_groups, "lanes_of_the_context": n_lanes,
                  # (lfg_capi.cpp, motion_run: with frames in flight a call whose lane's previous call found most sample blocks matched
                  #  launches 5/8 of the persistent workgroups the device holds WHILE ANOTHER LANE IS BUSY -- more units per workgroup,
                  #  room for the other lanes' kernels: +4 % frames/s; a call that has the device to itself, as in the pass below, takes
                  #  them all)
                  "prefilter_workgroups_launched_beside_other_lanes": (prefilter_groups * 5 // 8) if n_lanes > 1 else prefilter_groups,
                  "how": ("events around every stage call inside the timed regions (one frame at a time: nothing overlaps)" if n_lanes == 1 and args.workload != "scale" else
                          "a second pass, one call at a time on lane 0 of the SAME context as the timed regions, event-bracketed: the same "
                          f"work-unit plan (rim segments in {'4 parts' if rim_split == 4 else '4 parts, 8 at the top and bottom border' if rim_split == 48 else '8 parts'}) "
                          "as the step that `value` times, where the stages of neighbouring steps overlap and an event pair would time both; a context that runs "
                          "one frame at a time (--in-flight 1) plans for the longest unit instead (rim split 48) and its motion call is shorter: "
                          "profiles/r03_pipeline_one_lane_bench.json")}
    stage_ms = {}
    for name, sid in (("scale", capi.STAGE_SCALE), ("motion", capi.STAGE_MOTION), ("interpolate", capi.STAGE_INTERPOLATE)):
        ms, n = ctx.profile_get(sid)
        if n:
            stage_ms[name] = ms / n
    ctx.profile_enable(False)

    # what RCCL saw, and which card every rank drove (the driver checks its N against these)
    devices, rccl_ranks, verified = gather_ranks(dist if world > 1 else None, world, dev_index, ctx.comm_ranks(), verified)

    units_per_step = 1 if args.workload == "scale" else len(factors)
    in_res = args.workload == "pipeline_input_res"
    mw, mh = (w_in, h_in) if in_res else (w, h)        # resolution motion and interpolate run at
    value = world * steps * units_per_step / elapsed

    # ---- the same run, rank 0, one GPU: the two HBM-bound configurations and the other contents
    extras = {}

    def own_process(flag, key):
        """One of the labelled extras measured by `bench.py FLAG` in a process of its own (a context of its own on this GPU while this process
        waits): {...} of its JSON line, or {"skipped": why} -- an extra must not take the line down."""
        import subprocess
        try:
            sub = subprocess.run([sys.executable, os.path.abspath(__file__), flag, "--in-flight", str(n_lanes)], capture_output=True, text=True, timeout=300)
            last = [ln for ln in sub.stdout.strip().splitlines() if ln.startswith("{")]
            return json.loads(last[-1])[key] if last else {"skipped": "no output: " + sub.stderr[-300:]}
        except Exception as e:           # noqa: BLE001
            return {"skipped": repr(e)}
    if rank == 0 and world == 1 and not args.no_extras and args.workload == "pipeline" and args.input == "1080p" and args.semantics == "reference":
        n2 = 4000
        b_scale = algorithmic_bytes("scale", w_in, h_in, w, h)
        b_interp = algorithmic_bytes("interpolate", w_in, h_in, w, h) + 4 * w * h * (len(factors) - 1)
        t_s = timed(lambda k: ctx.scale(f_curr_in, f_curr4), n2)
        # scale + interpolate is measured on content where interpolate really SAMPLES: static frames (curr = prev, zero vectors:
        # every pixel fetches both frames -- 14 B/px moved).  Under the benchmark's pan the literal semantics
        # reject both samples of every pixel, the stage reads 16.6 MB of vectors, writes 33.2 MB of zeros, and a fraction
        # computed from its 116 MB of ALGORITHMIC bytes would flatter it by 2.3x: that figure is printed beside, on bytes moved.
        t_mv0 = torch.zeros((h, w, 2), dtype=torch.uint8, device=dev)
        f_mv0 = capi.Context.wrap(t_mv0.data_ptr(), w, h, capi.FORMAT_MV_S8X2)
        mv_pan = t_mv.cpu().numpy().view(np.int8)
        moved_pan = interpolate_bytes_moved(mv_pan, factors)
        def scale_interp_static(k):
            ctx.scale(f_prev_in, f_curr4)                             # curr = prev: the same pixels as f_prev4, in a buffer of their own
            interpolate_all(f_prev4, f_curr4, f_mv0)
        def scale_interp_pan(k):
            ctx.scale(f_curr_in, f_curr4)
            interpolate_all(f_prev4, f_curr4, f_mv)
        timed(scale_interp_static, 50)
        t_si = timed(scale_interp_static, n2)
        timed(scale_interp_pan, 50)
        t_sip = timed(scale_interp_pan, n2)
        # the interpolate call alone on both contents, event-bracketed on one lane (HIP events around the launch)
        def interp_alone(fm):
            ctx.profile_reset(); ctx.profile_enable(True)
            timed(lambda k: interpolate_all(f_prev4, f_curr4, fm), 400)
            ms, cnt = ctx.profile_get(capi.STAGE_INTERPOLATE)
            ctx.profile_enable(False)
            return ms / max(cnt, 1)
        if n_lanes > 1:
            ctx.lane_select(0)
        ctx.scale(f_prev_in, f_curr4)
        ms_i_static = interp_alone(f_mv0)
        ctx.scale(f_curr_in, f_curr4)
        ms_i_pan = interp_alone(f_mv)
        extras["scale_only"] = {"workload": "BASELINE config 2: 1080p->4K Lanczos only", "steps": n2, "frames_per_s": round(n2 / t_s, 1),
                                "us_per_step": round(t_s / n2 * 1e6, 3), "algorithmic_bytes": b_scale,
                                "hbm_gbs": round(b_scale * n2 / t_s / 1e9, 1), "hbm_frac": round(b_scale * n2 / t_s / 1e9 / HBM_PEAK_GBS, 4),
                                "how": "wall clock over back-to-back launches (includes the ~1.5 us between dependent kernels); "
                                       "rocprof's GPU-side duration of the kernel is in profiles/"}
        def frac(nbytes, seconds, n):
            return round(nbytes * n / seconds / 1e9 / HBM_PEAK_GBS, 4)
        extras["scale_interpolate"] = {"workload": "scale + interpolate (motion vectors given)",
                                       "content": "static: curr = prev, zero vectors -- every pixel samples both frames, bytes moved = algorithmic bytes",
                                       "steps": n2, "frames_per_s": round(n2 / t_si, 1),
                                       "us_per_step": round(t_si / n2 * 1e6, 3), "algorithmic_bytes": b_scale + b_interp, "bytes_moved": b_scale + b_interp,
                                       "hbm_gbs": round((b_scale + b_interp) * n2 / t_si / 1e9, 1),
                                       "hbm_frac": frac(b_scale + b_interp, t_si, n2),
                                       "on_the_headline_content": {"content": args.content, "us_per_step": round(t_sip / n2 * 1e6, 3), "algorithmic_bytes": b_scale + b_interp,
                                                                "bytes_moved": b_scale + moved_pan, "hbm_frac": frac(b_scale + moved_pan, t_sip, n2),
                                                                "note": "on the pan both samples of every pixel are rejected (SURVEY.md F5): interpolate reads the vectors and writes "
                                                                        "zeros; the fraction is of the bytes it MOVES, not of its 116 MB of algorithmic bytes"}}
        extras["interpolate_alone"] = {
            "how": "HIP events around 400 lfg_interpolate launches, one at a time on one lane; inputs warm (the same buffers every launch)",
            "static": {"avg_us": round(ms_i_static * 1e3, 3), "bytes_moved": b_interp, "hbm_frac": round(b_interp / (ms_i_static * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "every pixel samples both frames"},
            "headline_content": {"content": args.content, "avg_us": round(ms_i_pan * 1e3, 3), "algorithmic_bytes": b_interp, "bytes_moved": moved_pan,
                              "hbm_frac": round(moved_pan / (ms_i_pan * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "on the pan both samples are rejected (F5): fraction of the bytes moved"}}
        # ---- the same two configurations on TWO lanes of a second context (launch k + 1's dispatch and first-row wait overlap
        # launch k's tail: a throughput figure, labelled) and cache-COLD (every buffer of a step was last touched more than a
        # gigabyte of traffic ago, far beyond the 256 MiB Infinity Cache -- except the upscaled frame interpolate reads, which
        # the step's own scale has just written: that locality is the pipeline's, not the benchmark's)
        ctx2 = capi.Context(dev_index)
        ctx2.lanes(2)
        two = [empty_frame(w, h) for _ in range(2)]
        def scale_two(k):
            ctx2.lane_select(k & 1)
            ctx2.scale(f_curr_in, two[k & 1][1])
        timed(scale_two, 200)
        t_s2 = timed(scale_two, n2)
        extras["scale_only"]["two_lanes"] = {"frames_per_s": round(n2 / t_s2, 1), "us_per_step": round(t_s2 / n2 * 1e6, 3),
                                             "hbm_frac": round(b_scale * n2 / t_s2 / 1e9 / HBM_PEAK_GBS, 4),
                                             "how": "independent frames alternate between two streams of one context (lfg_lanes 2): throughput of "
                                                    "overlapping launches, not the duration of one"}
        n_sets = 12
        sets = []
        for i in range(n_sets):
            tin = t_curr_in.clone()
            sets.append((tin, capi.Context.wrap(tin.data_ptr(), w_in, h_in, capi.FORMAT_RGBA8), empty_frame(w, h), empty_frame(w, h),
                         empty_frame(w, h, capi.FORMAT_MV_S8X2), [empty_frame(w, h) for _ in factors]))
            sets[-1][3][0].copy_(t_prev4); sets[-1][4][0].zero_()          # (static content: the input is the previous frame's, zero vectors)
            sets[-1][0].copy_(t_prev_in)
        set_bytes = w_in * h_in * 4 + (2 + len(factors)) * w * h * 4 + w * h * 2
        def scale_cold(k):
            st = sets[k % n_sets]
            ctx.scale(st[1], st[2][1])
        def scale_interp_cold(k):
            st = sets[k % n_sets]
            ctx.scale(st[1], st[2][1])
            interpolate_all(st[3][1], st[2][1], st[4][1], [f for _, f in st[5]])
        timed(scale_cold, 2 * n_sets)
        t_sc = timed(scale_cold, n2)
        timed(scale_interp_cold, 2 * n_sets)
        t_sic = timed(scale_interp_cold, n2)
        extras["scale_only"]["cache_cold"] = {"frames_per_s": round(n2 / t_sc, 1), "us_per_step": round(t_sc / n2 * 1e6, 3),
                                              "hbm_frac": round(b_scale * n2 / t_sc / 1e9 / HBM_PEAK_GBS, 4),
                                              "how": f"{n_sets} distinct input and output frames in rotation ({n_sets * (w_in * h_in + w * h) * 4 >> 20} MiB between two uses of a buffer)"}
        extras["scale_only"]["cache"] = "warm: one input frame, resident in the Infinity Cache between launches (cache_cold beside it)"
        extras["scale_interpolate"]["cache_cold"] = {"frames_per_s": round(n2 / t_sic, 1), "us_per_step": round(t_sic / n2 * 1e6, 3),
                                                     "bytes_moved": b_scale + b_interp, "hbm_frac": round((b_scale + b_interp) * n2 / t_sic / 1e9 / HBM_PEAK_GBS, 4),
                                                     "how": f"{n_sets} distinct sets of input / upscaled / previous / vector / output frames in rotation "
                                                            f"({n_sets * set_bytes >> 20} MiB between two uses of a buffer); the upscaled frame interpolate "
                                                            "reads was written by the step's own scale"}
        extras["scale_interpolate"]["cache"] = "warm: the same buffers every step (cache_cold beside it)"
        del sets, two
        ctx2.close()
        sweep = {args.content: {"frames_per_s": round(value, 1), "motion_ms": round(stage_ms.get("motion", 0.0), 4)}}
        def sweep_case(label, content, mode):
            p_in, c_in = make_content(content, w_in, h_in, 0, 0)
            tp, fp = dev_frame(p_in)
            tc, fc = dev_frame(c_in)
            ctx.set_motion_mode(mode)
            ctx.scale(fp, f_prev4)
            n = 5 if mode == capi.MOTION_EXACT_ONLY else (20 if content in ("uncorrelated", "fade") else 100)
            timed(lambda k: pipeline_step(k, fc, n_lanes), 2 * n_lanes)
            t = timed(lambda k: pipeline_step(k, fc, n_lanes), n)                # frames/s: as `value`, frames in flight
            if n_lanes > 1:
                ctx.lane_select(0)
            ctx.profile_reset(); ctx.profile_enable(True)
            timed(lambda k: pipeline_step(k, fc, 1), max(2, n // 4))              # motion_ms: one lane, event-bracketed
            ms, cnt = ctx.profile_get(capi.STAGE_MOTION)
            ctx.profile_enable(False)
            entry = {"frames_per_s": round(n * len(factors) / t, 1), "motion_ms": round(ms / max(cnt, 1), 4), "steps": n}
            if mode == capi.MOTION_PREFILTERED:
                st = ctx.motion_last_stats()
                entry["fallback_tiles"] = st[1]
            sweep[label] = entry
        if not exact_only:
            for c in CONTENTS:
                if c != args.content:
                    sweep_case(c, c, capi.MOTION_PREFILTERED)
            # the pan under sensor noise of other amplitudes than the benchmark's +-2 levels (at the 1080p input): from +-3 on the lane's previous
            # call selects the persistent kernel's variant with the walks by SADs (include/linuxfg_hip.h: lfg_motion_last_variant)
            saved_amp = os.environ.get("LFG_BENCH_NOISE_AMP")
            for amp in (1, 3, 4, 6, 8, 12):
                os.environ["LFG_BENCH_NOISE_AMP"] = str(amp)
                label = "noisy, +-%d levels" % amp
                sweep_case(label, "noisy", capi.MOTION_PREFILTERED)
                sweep[label]["persistent_kernel_variant"] = ctx.motion_last_variant()
            if saved_amp is None:
                os.environ.pop("LFG_BENCH_NOISE_AMP", None)
            else:
                os.environ["LFG_BENCH_NOISE_AMP"] = saved_amp
            sweep_case("literal kernel only (any content)", "translated", capi.MOTION_EXACT_ONLY)
            ctx.set_motion_mode(capi.MOTION_PREFILTERED)
            ctx.scale(f_prev_in, f_prev4)
        worst = min((v["frames_per_s"] for k, v in sweep.items() if not k.startswith("literal")), default=None)
        # ---- the N > 1 step on this one GPU: with several GPUs the batch shares its previous frame, which arrives as an INPUT
        # frame per step and is upscaled by every rank -- two scales per step instead of one.  A scaling curve's N = 1 point has
        # to be THIS data flow (no collective: there is nobody to send to), not `value`.
        if not share_input:
            slots2 = [dev_frame(prev_in) for _ in range(2)]
            bufs2 = [(fc4, fmv, fouts, empty_frame(w, h)) for (fc4, fmv, fouts, _) in lane_bufs]
            bufs2f = [(a, b, c, d[1]) for (a, b, c, d) in bufs2]
            sh2 = sharding.SharedFrameBroadcaster(2, None, world_size=1, is_source=True)
            n3 = 400
            timed(lambda k: pipeline_step(k, f_curr_in, n_lanes, sh2, bufs2f, slots2), 4 * n_lanes)
            t_same = timed(lambda k: pipeline_step(k + 4 * n_lanes, f_curr_in, n_lanes, sh2, bufs2f, slots2), n3)
            if n_lanes > 1:
                ctx.lane_select(0)
            extras["same_dataflow_as_n_gt_1"] = {"frames_per_s": round(n3 * len(factors) / t_same, 1), "ms_per_step": round(t_same / n3 * 1e3, 5), "steps": n3,
                                                 "how": "the step every rank runs at N > 1 -- scale(shared previous INPUT frame) + scale(curr) + motion + interpolate -- "
                                                        "on this GPU, without the broadcast: the denominator a scaling curve over N should use"}
            del slots2, bufs2
            # (a process of its own, as a rank is: this one holds the headline context's streams as well, and with more streams than
            #  hardware queues -- GPU_MAX_HW_QUEUES -- two lanes share one and run in turn: 2,420 instead of 3,400 frames/s measured here)
            extras["with_communicator"] = own_process("--with-communicator", "with_communicator")
        # ---- the opt-in intended semantics (lfg_set_semantics: vectors displace by pixels, ties go to the shortest vector) -- the only
        # mode whose generated frames mean anything -- on the pan and on moving objects, frames in flight as `value`
        intended = {}
        ctx.set_semantics(capi.SEMANTICS_INTENDED)
        for label in ("translated", "objects"):
            p_in, c_in = make_content(label, w_in, h_in, 0, 0)
            tp, fp = dev_frame(p_in)
            tc, fc = dev_frame(c_in)
            if n_lanes > 1:
                ctx.lane_select(0)
            ctx.scale(fp, f_prev4)
            timed(lambda k: pipeline_step(k, fc, n_lanes), 4 * n_lanes)
            n_i = 200
            t_i = timed(lambda k: pipeline_step(k, fc, n_lanes), n_i)
            if n_lanes > 1:
                ctx.lane_select(0)
            ctx.profile_reset(); ctx.profile_enable(True)
            timed(lambda k: pipeline_step(k, fc, 1), 40)
            per = {nm: round(ctx.profile_get(sid)[0] / max(ctx.profile_get(sid)[1], 1), 4)
                   for nm, sid in (("scale", capi.STAGE_SCALE), ("motion", capi.STAGE_MOTION), ("interpolate", capi.STAGE_INTERPOLATE))}
            ctx.profile_enable(False)
            moved = interpolate_bytes_moved(lane_tensors[0][0].cpu().numpy().view(np.int8), factors, intended=True)
            used, tiles, left = ctx.motion_lean_stats()
            intended[label] = {"frames_per_s": round(n_i * len(factors) / t_i, 1), "ms_per_step": round(t_i / n_i * 1e3, 5), "steps": n_i,
                               "stage_ms_one_call_at_a_time": per, "interpolate_bytes_moved": moved,
                               "lean_kernel": {"used": used, "tiles": tiles, "tiles_left": left}}
        ctx.set_semantics(capi.SEMANTICS_REFERENCE)
        if n_lanes > 1:
            ctx.lane_select(0)
        ctx.scale(f_prev_in, f_prev4)
        intended["how"] = ("lfg_set_semantics(INTENDED) on the headline context, same step and frames in flight as `value`; under these semantics interpolate "
                           "samples displaced texels on every content (no rejected samples), and ties resolve to the shortest vector")
        extras["intended_semantics"] = intended
        # (contexts of their own in processes of their own, like with_communicator above: beside the headline context's streams theirs share
        #  hardware queues -- config 5 measured 3 % slower inside this process than alone)
        extras["stream"] = own_process("--stream", "stream")
        extras["pcie_inclusive"] = measure_pcie_inclusive(torch, capi, ctx, dev, w_in, h_in, w, h)
        extras["config5"] = own_process("--config5", "config5")
        extras["content_sweep"] = {"frames_per_s_by_content": sweep, "worst_case_frames_per_s": worst,
                                   "note": "same kernels, same results discipline (bit-exact vectors on every content); the motion "
                                           "stage's run time depends on how much of the frame has an exact or near match inside the "
                                           "search range: `value` is on the SURVEY-prescribed pure pan, its best case bar `static`"}

    if rank == 0:
        from types import SimpleNamespace
        interp_moved = None
        if "interpolate" in stage_ms:
            interp_moved = interpolate_bytes_moved((t_mv_in if in_res else t_mv).cpu().numpy().view(np.int8), factors, intended=args.semantics == "intended")
        line = assemble_line(SimpleNamespace(
            args=args, world=world, stage_ms=stage_ms, factors=factors, w_in=w_in, h_in=h_in, w=w, h=h, mw=mw, mh=mh, in_res=in_res,
            share_input=share_input, steps=steps, warmup=warmup, elapsed=elapsed, regions=regions, value=value, exact_only=exact_only,
            motion_stats=motion_stats, n_lanes=n_lanes, devices=devices, rccl_ranks=rccl_ranks, fused_mi=fused_mi, extras=extras,
            stage_pass=stage_pass, interp_moved=interp_moved, verified=verified,
            cpu_baseline=(None if args.no_cpu_baseline else (lambda: cpu_baseline(w_in, h_in, w, h, factors, args.workload)))))
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
