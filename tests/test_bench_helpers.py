"""bench.py's host-side helpers (no GPU): the per-rank motion of the multi-GPU workload and the readers of the committed
rocprofv3 counter summaries that fill `roofline.traffic` and `roofline.executed`."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_rank_moves_inside_the_search_range():
    """A rank whose frames move by more than 16 output pixels has no match anywhere: its motion stage would search in
    full (ten times slower) and, timed by the slowest rank, wreck the multi-GPU figure."""
    b = _bench()
    seen = set()
    for rank in range(64):
        dx, dy = b.rank_motion(rank)
        assert abs(2 * dx) <= 16 and abs(2 * dy) <= 16, (rank, dx, dy)
        if rank < 8:
            seen.add((dx, dy))
    assert b.rank_motion(0) == (3, -2)                       # the single-GPU workload of BASELINE config 3
    assert len(seen) == 8                                    # eight GPUs, eight different motions


SAMPLE_TRAFFIC = """# lib_sha16 0123456789abcdef
# steps 10
kernel lfg::interpolate_kernel launches_per_step 1 mean_us 20.27 FETCH_SIZE 8185 WRITE_SIZE 32400
kernel lfg::motion_prefilter_kernel launches_per_step 2 mean_us 319.6 FETCH_SIZE 116200 WRITE_SIZE 162200
kernel lfg::motion_resolve_kernel launches_per_step 1 mean_us 48.4 FETCH_SIZE 11000 WRITE_SIZE 388
kernel lfg::scale_2x_kernel launches_per_step 1 mean_us 12.9 FETCH_SIZE 6616 WRITE_SIZE 32630
per_step FETCH_SIZE 258201 WRITE_SIZE 389818
"""
SAMPLE_SQ = """# lib_sha16 0123456789abcdef
# steps 10
kernel lfg::motion_prefilter_kernel launches_per_step 2 mean_us 320.0 SQ_INSTS_LDS 4450000 SQ_INSTS_SALU 9000000 SQ_INSTS_VALU 90250000 SQ_WAVES 40000
kernel lfg::scale_2x_kernel launches_per_step 1 mean_us 12.9 SQ_INSTS_LDS 280000 SQ_INSTS_SALU 500000 SQ_INSTS_VALU 3270000 SQ_WAVES 4352
per_step SQ_INSTS_VALU 183770000
"""


def test_counter_tables_are_read_back_only_for_the_library_that_made_them(tmp_path, monkeypatch):
    """`roofline.traffic` and `dominant_stage.executed` come from committed counter tables (tools/pmc_per_step.py); a
    format drift must not turn them into None unnoticed, and a table taken with another build must not be quoted."""
    b = _bench()
    (tmp_path / "profiles").mkdir()
    (tmp_path / "profiles" / "r02_hbm_traffic_pmc.txt").write_text(SAMPLE_TRAFFIC)
    (tmp_path / "profiles" / "r02_sq_counters.txt").write_text(SAMPLE_SQ)
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    monkeypatch.setattr(b, "library_sha16", lambda: "0123456789abcdef")
    traffic, src = b.pmc_traffic("lfg::")
    assert src == "profiles/r02_hbm_traffic_pmc.txt"
    want = (8185 + 32400 + 2 * (116200 + 162200) + 11000 + 388 + 6616 + 32630) * 1024
    assert traffic == want
    assert b.pmc_traffic("lfg::scale_2x")[0] == (6616 + 32630) * 1024
    ex = b.pmc_executed("lfg::motion_")
    assert ex["valu_wave_instructions"] == 2 * 90250000 and ex["kernel_us_per_step_in_that_pass"] == 640.0
    # three stated conventions: 2 cycles per instruction (SIMD-32, two waves resident), the same cost-weighted (no weights file
    # for this made-up library: None), and round 4's 4 cycles (one wave alone)
    assert abs(ex["valu_issue_utilisation_simd32_2_cycles"] - 2 * 90250000 * 2 / (640e-6 * 2.4e9 * 1024)) < 1e-3
    assert abs(ex["valu_issue_utilisation_one_wave_alone_4_cycles"] - 2 * 90250000 * 4 / (640e-6 * 2.4e9 * 1024)) < 1e-3
    assert ex["valu_issue_utilisation_cost_weighted"] is None
    monkeypatch.setattr(b, "library_sha16", lambda: "ffffffffffffffff")       # another build: nothing is quoted
    t, why = b.pmc_traffic("lfg::")
    assert t is None and "not quoted" in why
    assert b.pmc_executed("lfg::motion_") is None


def test_contents_are_deterministic_and_distinct():
    b = _bench()
    frames = {c: b.make_content(c, 192, 108, 0, 0) for c in b.CONTENTS}
    again = b.make_content("occluded", 192, 108, 0, 0)
    assert (frames["occluded"][1] == again[1]).all()
    assert (frames["static"][0] == frames["static"][1]).all()
    cur = [frames[c][1].tobytes() for c in b.CONTENTS]
    assert len(set(cur)) == len(cur)


def test_gpus_n_starts_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` run bare (the driver's command shape) starts N ranks itself: a fresh child under
    torch.distributed.run with the same arguments, rendezvous on 127.0.0.1 -- never an exec, never a GPU call in the parent --
    and fewer devices than ranks is refused with a clear message before anything is launched."""
    import sys
    import types
    b = _bench()
    cmd = b.launch_command(4, ["--gpus", "4", "--steps", "5", "--warmup", "2"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "5", "--warmup", "2"]

    calls = []
    def fake_run(argv, **kw):
        calls.append((argv, kw))
        return types.SimpleNamespace(returncode=7)
    import subprocess
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("LFG_BENCH_SHARE_GPU", raising=False)
    monkeypatch.setattr(subprocess, "run", fake_run)
    args = types.SimpleNamespace(gpus=4)
    monkeypatch.setattr(b, "visible_devices", lambda: 8)           # an 8-GPU node: four ranks are started, their exit code comes back
    assert b.spawn_ranks(args, ["--gpus", "4", "--steps", "5"]) == 7
    assert len(calls) == 1 and calls[0][0][:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert calls[0][0][-4:] == ["--gpus", "4", "--steps", "5"]
    assert calls[0][1]["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"      # dmabuf IPC: RCCL needs it on this pool
    calls.clear()
    monkeypatch.setattr(b, "visible_devices", lambda: 1)           # the one-GPU box: refused, nothing launched
    assert b.spawn_ranks(args, ["--gpus", "4"]) == 2
    assert calls == []


def test_main_spawns_before_it_imports_torch(monkeypatch):
    """main() hands over to the launcher before `import torch` (the parent must never initialise the GPU)."""
    import sys
    b = _bench()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    seen = {}
    def fake_spawn(args, argv):
        seen["gpus"], seen["argv"] = args.gpus, list(argv)
        return 5
    monkeypatch.setattr(b, "spawn_ranks", fake_spawn)
    import pytest
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 5 and seen == {"gpus": 2, "argv": ["--gpus", "2", "--steps", "3"]}


def test_interpolate_bytes_moved_counts_only_the_samples_taken():
    """Under the literal semantics a displaced sample outside [0,1]^2 is vec4(0) without a fetch: the bytes the
    stage moves depend on the vectors.  Zero vectors: 14 B/px; the benchmark's pan: 6 B/px; half and half in between."""
    import numpy as np
    b = _bench()
    w, h = 64, 32
    zero = np.zeros((h, w, 2), np.int8)
    assert b.interpolate_bytes_moved(zero, [0.5]) == 14 * w * h
    assert b.interpolate_bytes_moved(zero, [0.25, 0.5, 0.75]) == (2 + 8 + 12) * w * h
    pan = np.zeros((h, w, 2), np.int8); pan[...] = (-6, 4)
    assert b.interpolate_bytes_moved(pan, [0.5]) == 6 * w * h
    half = pan.copy(); half[:, : w // 2] = 0
    assert b.interpolate_bytes_moved(half, [0.5]) == 6 * w * h + 8 * (w // 2) * h
    one = np.zeros((h, w, 2), np.int8); one[..., 0] = 1        # uv + 0.5: curr's sample stays inside for the left half, prev's for the right
    assert b.interpolate_bytes_moved(one, [0.5]) == 6 * w * h + 4 * w * h


def _line_worker(rank, world, port, q):
    import types
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = _bench()
        verified = {"ok": rank == 0 or True, "step": 44, "vectors_vs_literal_kernel": {"pixels": 10, "differing": 0}}
        devices, rccl, verified = b.gather_ranks(dist, world, rank, world, verified)
        if rank == 0:
            args = types.SimpleNamespace(workload="pipeline", input="1080p", content="translated", no_cpu_baseline=False, semantics="reference")
            r = types.SimpleNamespace(
                args=args, world=world, stage_ms={"scale": 0.012, "motion": 0.33, "interpolate": 0.017}, factors=[0.5],
                w_in=1920, h_in=1080, w=3840, h=2160, mw=3840, mh=2160, in_res=False, share_input=True, steps=20, warmup=5,
                elapsed=0.0066, regions=[0.0066, 0.0067, 0.0065], value=world * 20 / 0.0066, exact_only=False,
                motion_stats=(2040, 0, 0.0), n_lanes=3, devices=devices, rccl_ranks=rccl, fused_mi=False, extras={},
                stage_pass={"how": "stub"}, interp_moved=6 * 3840 * 2160, verified=verified,
                cpu_baseline=lambda: (_ for _ in ()).throw(AssertionError("the CPU baseline is an N = 1 matter")))
            q.put(b.assemble_line(r))
        else:
            q.put(None)
    finally:
        dist.destroy_process_group()


def test_the_line_of_an_n_gt_1_run_has_every_key():
    """The line rank 0 prints at N > 1, assembled on two CPU ranks over gloo from stub measurements: the contract's keys, the
    `roofline` object, `cpu_baseline: null` (it is timed at N = 1 only), every rank's device and check gathered in rank order."""
    import json
    import socket
    import torch.multiprocessing as mp
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_line_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    line = next(g for g in got if g is not None)
    json.dumps(line)                                         # serialisable as it stands
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "stages", "verified", "rccl_ranks", "library_sha16"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["cpu_baseline"] is None
    assert line["config"]["devices"] == [0, 1] and line["rccl_ranks"] == 2
    rf = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1
    # two scales per step at N > 1 (the shared previous frame is upscaled by every rank): the path's bytes say so
    assert rf["algorithmic_bytes_per_step"] == 2 * 41472000 + 82944000 + 116121600
    assert line["stages"]["interpolate"]["bytes_moved"] == 6 * 3840 * 2160
    assert line["verified"]["ok"] is True and line["verified"]["per_rank_ok"] == [True, True]
