"""The C++ host mirror (linux-fg_amd/host: Scaler / FrameManager / HipContext, built as lfg_host)
driven end to end on the GPU and checked against the oracle: ProcessFrame's presentation order
(real, then [generated, real] per further input frame), the upscaled frames (+-1 LSB) and the
generated frames (exact, from the device's own upscaled frames)."""
import json
import os
import subprocess

import numpy as np
import pytest

from linux_fg_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "linux-fg_amd", "lfg_host")


@pytest.fixture(scope="module")
def host_binary():
    if not os.path.exists(HOST):
        import __graft_entry__ as entry
        entry.build()
    return HOST


def test_process_frame_sequence_matches_oracle(host_binary, oracle, tmp_path):
    w, h, n = 64, 36, 3
    out = subprocess.run([host_binary, "--input-width", str(w), "--input-height", str(h), "--output-width", str(2 * w),
                          "--output-height", str(2 * h), "--frames", str(n), "--dump-dir", str(tmp_path), "--quiet", "2"],
                         capture_output=True, text=True, check=True)
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["presented"] == 2 * n - 1 and stats["interpolated"] == n - 1
    files = sorted(os.listdir(tmp_path))
    kinds = [f.split("_")[2] for f in files]
    assert kinds == ["real", "interp", "real", "interp", "real"]         # presentation order
    frames = [np.fromfile(os.path.join(tmp_path, f), np.uint8).reshape(2 * h, 2 * w, 4) for f in files]
    # inputs as the C++ SyntheticCapture generates them (stream 2)
    seed = synth.BASE_SEED + 2
    inputs = [synth.make_prev(w, h, seed)]
    for k in range(1, n):
        inputs.append(synth.translate(inputs[-1], (3, -2), seed + k))
    reals = [frames[0], frames[2], frames[4]]
    for got, src in zip(reals, inputs):
        d = np.abs(got.astype(np.int16) - oracle.scale(src, 2 * w, 2 * h).astype(np.int16))
        assert d.max() <= 1
    for i, gen in enumerate([frames[1], frames[3]]):
        prev_up, curr_up = reals[i], reals[i + 1]
        mv = oracle.motion(prev_up, curr_up, 8, 16.0)
        assert (gen == oracle.interpolate(prev_up, curr_up, mv, 0.5)).all()


def test_cadence_60_to_240_sequence_matches_oracle(host_binary, oracle, tmp_path):
    """lfg_host --factors 0.25,0.5,0.75 (BASELINE config 5's cadence): per further input frame
    three generated frames then the real one -- real, [t=1/4, t=1/2, t=3/4, real] ... -- motion once per pair, every
    generated frame equal to the oracle's for its factor, in both presentation modes."""
    w, h, n = 64, 36, 3
    factors = [0.25, 0.5, 0.75]
    for extra in ([], ["--sync-present"]):
        d = tmp_path / ("cadence" + "".join(extra))
        d.mkdir()
        stats = run_host(["--input-width", str(w), "--input-height", str(h), "--output-width", str(2 * w), "--output-height", str(2 * h),
                          "--frames", str(n), "--dump-dir", str(d), "--quiet", "--factors", "0.25,0.5,0.75", "2"] + extra)
        assert stats["presented"] == n + 3 * (n - 1) and stats["interpolated"] == 3 * (n - 1)
        files = sorted(os.listdir(d))
        assert [f.split("_")[2] for f in files] == ["real"] + ["interp", "interp", "interp", "real"] * (n - 1)
        frames = [np.fromfile(d / f, np.uint8).reshape(2 * h, 2 * w, 4) for f in files]
        reals = frames[0::4]
        seed = synth.BASE_SEED + 2
        inputs = [synth.make_prev(w, h, seed)]
        for k in range(1, n):
            inputs.append(synth.translate(inputs[-1], (3, -2), seed + k))
        for got, src in zip(reals, inputs):
            assert np.abs(got.astype(np.int16) - oracle.scale(src, 2 * w, 2 * h).astype(np.int16)).max() <= 1
        for i in range(n - 1):
            mv = oracle.motion(reals[i], reals[i + 1], 8, 16.0)
            for j, t in enumerate(factors):
                assert (frames[4 * i + 1 + j] == oracle.interpolate(reals[i], reals[i + 1], mv, t)).all(), (i, t)


def test_shared_previous_batch_mode_one_rank(host_binary, oracle, tmp_path):
    """lfg_host --ranks 1 --rank 0 --comm-file F (BASELINE config 4's data flow in the C++ host, on the one GPU of this
    box): the communicator id travels through the file, call k's previous frame is frame k of stream 0 -- captured on
    rank 0, broadcast through lfg_broadcast_frame one call ahead, upscaled on the rank -- and its current frame is
    frame k of the rank's own stream.  Every call presents [generated, real]; each generated frame equals the
    oracle's for that pair.  (More ranks need more GPUs: RCCL wants one device per rank.)"""
    w, h, n = 64, 36, 3
    d = tmp_path / "shared"
    d.mkdir()
    # a file left behind by an earlier run under the same name (another run's nonce, garbage for an id): rank 0 replaces it,
    # nobody joins it
    (tmp_path / "comm.id").write_bytes(b"LFGCOMM1" + (12345).to_bytes(8, "little") + bytes(128))
    stats = run_host(["--input-width", str(w), "--input-height", str(h), "--output-width", str(2 * w), "--output-height", str(2 * h),
                      "--frames", str(n), "--dump-dir", str(d), "--quiet", "--ranks", "1", "--rank", "0",
                      "--comm-file", str(tmp_path / "comm.id"), "--comm-nonce", "777"])
    assert stats["presented"] == 2 * n and stats["interpolated"] == n
    assert not os.path.exists(tmp_path / "comm.id")          # every rank has joined: rank 0 removed the id, nothing to go stale
    files = sorted(os.listdir(d))
    assert [f.split("_")[2] for f in files] == ["interp", "real"] * n
    frames = [np.fromfile(d / f, np.uint8).reshape(2 * h, 2 * w, 4) for f in files]

    def stream(idx):
        seed = synth.BASE_SEED + idx
        out = [synth.make_prev(w, h, seed)]
        for k in range(1, n):
            out.append(synth.translate(out[-1], (3, -2), seed + k))
        return out
    shared, own = stream(0), stream(1)
    for k in range(n):
        real = frames[2 * k + 1]
        assert np.abs(real.astype(np.int16) - oracle.scale(own[k], 2 * w, 2 * h).astype(np.int16)).max() <= 1
        prev_up = oracle.scale(shared[k], 2 * w, 2 * h)              # the device's own upscale may differ by an LSB: take
        mv = oracle.motion(prev_up, real, 8, 16.0)                  # the vectors and the blend from what it presented
        gen = oracle.interpolate(prev_up, real, mv, 0.5)
        diff = np.abs(frames[2 * k].astype(np.int16) - gen.astype(np.int16))
        assert diff.max() <= 1 or (diff > 1).mean() < 0.02, k        # (an LSB in prev_up can move a blend by one, rarely a vector)


def test_no_interpolation_and_aspect_ratio(host_binary):
    out = subprocess.run([host_binary, "--input-width", "80", "--input-height", "40", "--output-height", "100",
                          "--no-interpolation", "--frames", "4", "--quiet"], capture_output=True, text=True, check=True)
    stats = json.loads(out.stdout.strip().splitlines()[-1])
    assert stats["presented"] == 4 and stats["interpolated"] == 0      # output width derived: 200 (src/main.cpp:76-90)


def run_host(args):
    out = subprocess.run([HOST] + args, capture_output=True, text=True, check=True, timeout=300)
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_several_ranks_need_a_run_nonce(host_binary, tmp_path):
    """Without a nonce of its run a rank > 0 could join an earlier run's communicator id (nonce 0 in a stale file) and wait in
    ncclCommInitRank until the watchdog: more than one rank without --comm-nonce / LFG_COMM_NONCE is refused at once."""
    env = {k: v for k, v in os.environ.items() if k != "LFG_COMM_NONCE"}
    out = subprocess.run([HOST, "--input-width", "64", "--input-height", "36", "--output-width", "128", "--output-height", "72",
                          "--frames", "1", "--quiet", "--ranks", "2", "--rank", "1", "--comm-file", str(tmp_path / "comm.id")],
                         capture_output=True, text=True, timeout=60, env=env)
    assert out.returncode != 0
    assert "nonce" in (out.stderr + out.stdout)


def _presented(tmp_path, w, h, mode_args, frames=4):
    d = tmp_path / ("dump_" + ("_".join(a.strip("-") for a in mode_args) if mode_args else "default") + f"_{frames}")
    d.mkdir()
    info = run_host(["--input-width", str(w), "--input-height", str(h), "--output-width", str(2 * w),
                     "--frames", str(frames), "--quiet", "--dump-dir", str(d)] + mode_args)
    files = sorted(os.listdir(d))
    return info, files, [np.fromfile(d / f, np.uint8) for f in files]


def test_pipelined_presentation_matches_synchronous(host_binary, tmp_path):
    """Pipelined read-back (copy stream, frames presented one call later) must present the same frames in
    the same order as the reference's wait-per-call behaviour."""
    w, h = 96, 64
    a_info, a_files, a = _presented(tmp_path, w, h, [])
    b_info, b_files, b = _presented(tmp_path, w, h, ["--sync-present"])
    assert a_info["pipelined"] is True and b_info["pipelined"] is False
    assert a_files == b_files and a_info["presented"] == b_info["presented"] == 7
    assert a_info["checksum"] == b_info["checksum"]
    for x, y in zip(a, b):
        assert (x == y).all()


def test_frames_in_flight_present_the_same_stream(host_binary, tmp_path):
    """--in-flight N (lanes of the C-ABI: call k's upload, kernels and read-backs on lane k % N, the upscaled frames in a
    rotation of N + 1 buffers): the presented stream is the one-frame-at-a-time stream, frame for frame, for one and for
    three generated frames per pair."""
    w, h, n = 96, 64, 8
    for extra in ([], ["--factors", "0.25,0.5,0.75"]):
        ref_info, ref_files, ref = _presented(tmp_path, w, h, ["--in-flight", "1"] + extra, frames=n)
        assert ref_info["presented"] == (n - 1) * (1 + (3 if extra else 1)) + 1
        for lanes in ("2", "3"):
            info, files, got = _presented(tmp_path, w, h, ["--in-flight", lanes] + extra, frames=n)
            assert files == ref_files and info["presented"] == ref_info["presented"] and info["checksum"] == ref_info["checksum"]
            for x, y in zip(got, ref):
                assert (x == y).all()


def test_raw_file_source_and_sink(host_binary, tmp_path):
    """--input-raw / --output-raw: the headless stand-ins for capture and display.  Frames fed from a file must
    give the same presented stream as the built-in synthetic source producing the same frames."""
    w, h, n = 96, 64, 3
    frames = [synth.make_prev(w, h, seed=synth.BASE_SEED)]
    for k in range(1, n):
        frames.append(synth.translate(frames[-1], (3, -2), seed=synth.BASE_SEED + k))
    src = tmp_path / "in.rgba"
    np.concatenate([f.reshape(-1) for f in frames]).tofile(src)
    out = tmp_path / "out.rgba"
    info = run_host(["--input-width", str(w), "--input-height", str(h), "--output-width", str(2 * w), "--frames", str(n),
                     "--quiet", "--input-raw", str(src), "--output-raw", str(out)])
    ref_info, _, ref = _presented(tmp_path, w, h, [], frames=n)
    assert info["presented"] == ref_info["presented"] == 2 * n - 1
    got = np.fromfile(out, np.uint8)
    assert got.size == sum(r.size for r in ref)
    assert (got == np.concatenate(ref)).all()
    # one frame too many: the source runs dry and the tool reports it
    p = subprocess.run([HOST, "--input-width", str(w), "--input-height", str(h), "--frames", str(n + 1), "--quiet",
                        "--input-raw", str(src)], capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "end of input" in (p.stdout + p.stderr)
