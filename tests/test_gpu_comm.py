"""The C-ABI's communicator calls on the GPU (include/linuxfg_hip.h: lfg_comm_*, lfg_broadcast_frame), one rank: RCCL is
opened at run time, a one-rank communicator is created, a frame is broadcast on the communication stream next to kernels
on the compute stream, and every refusal the header promises is checked.  More ranks need more GPUs (RCCL wants one device
per rank): the driver's multi-GPU bench runs that; the scheduling around these calls is tested over gloo in
tests/test_dist_gloo.py."""
import numpy as np
import pytest

from linux_fg_amd import synth

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_broadcast_and_refusals():
    from linux_fg_amd import capi
    cid = capi.Context.comm_unique_id()
    assert len(cid) == capi.COMM_ID_BYTES and any(cid)
    with capi.Context(0) as ctx:
        assert ctx.lib.lfg_comm_ranks(ctx.h) == 0 and ctx.lib.lfg_comm_rank(ctx.h) == -1
        f = ctx.frame_from(synth.make_prev(320, 180, seed=5))
        with pytest.raises(capi.LfgError, match="no communicator"):
            ctx.broadcast_frame(f, 0)
        with pytest.raises(capi.LfgError, match="rank < nranks"):
            ctx.comm_init(2, 2, cid)
        ctx.comm_init(1, 0, cid)
        assert ctx.lib.lfg_comm_ranks(ctx.h) == 1 and ctx.lib.lfg_comm_rank(ctx.h) == 0
        with pytest.raises(capi.LfgError, match="already has a communicator"):
            ctx.comm_init(1, 0, cid)
        # a broadcast between kernels: scale reads f before and after; the pixels are those of the root (this rank)
        up1, up2 = ctx.create_frame(640, 360), ctx.create_frame(640, 360)
        ctx.scale(f, up1)
        ctx.broadcast_frame(f, 0)
        ctx.comm_wait()
        ctx.scale(f, up2)
        ctx.comm_wait()                                       # nothing pending: a no-op
        assert (ctx.download(f) == synth.make_prev(320, 180, seed=5)).all()
        assert (ctx.download(up1) == ctx.download(up2)).all()
        with pytest.raises(capi.LfgError, match="root out of range"):
            ctx.broadcast_frame(f, 1)
        wide = ctx.create_frame(330, 180)
        view = capi.Context.wrap(wide.data, 320, 180, capi.FORMAT_RGBA8, pitch=330 * 4)
        with pytest.raises(capi.LfgError, match="tightly packed"):
            ctx.broadcast_frame(view, 0)
        mv = ctx.frame_from(np.zeros((180, 320, 2), np.int8), capi.FORMAT_MV_S8X2)
        ctx.broadcast_frame(mv, 0)                            # motion-vector frames travel the same way
        ctx.comm_wait()
        ctx.comm_destroy()
        ctx.comm_destroy()                                    # idempotent
        with pytest.raises(capi.LfgError, match="no communicator"):
            ctx.comm_wait()
        ctx.comm_init(1, 0, capi.Context.comm_unique_id())    # a context can get a new communicator after destroying one
        ctx.broadcast_frame(f, 0)
        ctx.comm_wait()
        ctx.sync()
        # lfg_context_destroy tears the communicator down


def test_shared_frame_broadcast_with_frames_in_flight():
    """bench.py's multi-GPU step with lanes, on one rank: the double-buffered broadcast of the shared previous frame
    (SharedFrameBroadcaster over CapiTransport, a one-rank communicator standing in for the node's) while step k runs on
    lane k % n.  The lane's wait for the previous step's upscales is also what frees the slot the next frame lands in:
    every step must see ITS frame although the other lane is still busy with the step before."""
    from linux_fg_amd import capi, sharding
    w, h, steps = 320, 180, 7
    frames = [synth.make_prev(w, h, seed=900 + k) for k in range(steps + 1)]
    with capi.Context(0) as ctx:
        ctx.comm_init(1, 0, capi.Context.comm_unique_id())
        want = []
        for k in range(steps):                                # one lane, no broadcast: what every step has to produce
            f = ctx.frame_from(frames[k])
            up = ctx.create_frame(2 * w, 2 * h)
            ctx.scale(f, up)
            want.append(ctx.download(up))
            ctx.destroy_frame(f); ctx.destroy_frame(up)
        for lanes in (2, 3):
            ctx.lanes(lanes)
            slots = [ctx.create_frame(w, h) for _ in range(2)]
            ups = [ctx.create_frame(2 * w, 2 * h) for _ in range(steps)]
            mvs = [ctx.create_frame(2 * w, 2 * h, capi.FORMAT_MV_S8X2) for _ in range(lanes)]
            pinned = [ctx.staging_create(w * h * 4) for _ in range(steps + 1)]      # (asynchronous uploads: nothing here waits)
            for a, f in zip(pinned, frames):
                a[:] = f.reshape(-1)
            for lane_only in (False, True):                   # lfg_broadcast_frame, and lfg_broadcast_frame_lane: the step's lane_wait is what makes it safe
                bcast = sharding.SharedFrameBroadcaster(
                    2, sharding.CapiTransport(ctx, slots, src=0, behind_selected_lane_only=lane_only), world_size=2, is_source=True,
                    refill=lambda step, slot: ctx.upload_async(slots[slot], pinned[step]))
                ctx.lane_select(0)
                bcast.start(0)
                for k in range(steps):
                    ctx.lane_select(k % lanes)
                    ctx.lane_wait((k - 1) % lanes)
                    slot = bcast.acquire(k)
                    ctx.scale(slots[slot], ups[k])
                    ctx.lane_mark()
                    ctx.motion(ups[k], ups[k], mvs[k % lanes], 8, 16.0)      # something long on this lane after its upscale
                bcast.drain()
                ctx.sync()
                for k in range(steps):
                    assert (ctx.download(ups[k]) == want[k]).all(), (lanes, lane_only, k)
                    ctx.upload(ups[k], np.zeros((2 * h, 2 * w, 4), np.uint8))
            ctx.lane_select(0)
            for f in slots + ups + mvs:
                ctx.destroy_frame(f)
            for a in pinned:
                ctx.staging_destroy(a)
        ctx.lanes(1)


def test_broadcast_is_ordered_after_every_lane():
    """lfg_broadcast_frame is ordered after everything enqueued so far on EVERY lane's stream (include/linuxfg_hip.h), not
    after the selected lane only: a frame that another lane is still going to read must not be overwritten by what
    follows the broadcast.  Lane 0 queues something long and then an upscale that reads F; lane 1 -- with no lfg_lane_wait
    anywhere -- broadcasts F, waits for the broadcast and uploads new content into F.  The upscale has to see the OLD
    content.  (One rank: the broadcast moves no data, its ordering is what is tested.  A receiver's slot being overwritten
    by the broadcast itself needs a second GPU and cannot be seen here.)"""
    from linux_fg_amd import capi
    w, h = 320, 180
    old, new = synth.make_prev(w, h, seed=4101), synth.make_prev(w, h, seed=4102)
    big_a, big_b = synth.make_uncorrelated_pair(1280, 720, stream=3)      # a motion call of several milliseconds
    with capi.Context(0) as ctx:
        ctx.comm_init(1, 0, capi.Context.comm_unique_id())
        f = ctx.frame_from(old)
        up = ctx.create_frame(2 * w, 2 * h)
        ctx.scale(f, up)
        want = ctx.download(up).copy()
        ctx.lanes(2)
        fa, fb = ctx.frame_from(big_a), ctx.frame_from(big_b)
        mv = ctx.create_frame(1280, 720, capi.FORMAT_MV_S8X2)
        pinned = ctx.staging_create(w * h * 4)
        pinned[:] = new.reshape(-1)
        for attempt in range(6):
            ctx.lane_select(0)
            ctx.upload(f, old)
            ctx.sync()
            ctx.motion(fa, fb, mv, 8, 16.0)                   # lane 0 is busy ...
            ctx.scale(f, up)                                  # ... and reads F only afterwards
            ctx.lane_select(1)
            ctx.broadcast_frame(f, 0)
            ctx.comm_wait()
            ctx.upload_async(f, pinned)                       # F changes as soon as lane 1 gets there
            ctx.sync()
            assert (ctx.download(up) == want).all(), attempt
            assert (ctx.download(f) == new).all(), attempt
        ctx.lane_select(0)
        ctx.staging_destroy(pinned)
        for x in (f, up, fa, fb, mv):
            ctx.destroy_frame(x)
        ctx.lanes(1)


def _probe_behind_a_full_grid(capi, inputs, rounds=3):
    """Lane 0 runs a motion call that keeps the persistent kernel busy for a millisecond or two (a pan under sensor noise of +-8
    levels), a probe of RCCL's device kernel's footprint is issued like a broadcast (it becomes ready when lane 0's call has
    finished), and lane 1 is given a call on uncorrelated content at once (a persistent launch of 7 ms) -- whichever way the two
    launches share the chip at first, lane 1's holds all of it for milliseconds once lane 0's has run out of work.  Returns, per
    round, the probe's DEVICE time from ready to done (lfg_comm_probe_ms: 50 us of its own) and the host's milliseconds from
    the probe's end to lane 1's."""
    import time
    out = []
    with capi.Context(0) as ctx:
        ctx.lanes(2)
        ctx.comm_init(1, 0, capi.Context.comm_unique_id())
        reserved = ctx.comm_reserved_cus()
        h, w = inputs[0][0].shape[:2]
        frames = []
        for pair in inputs:                                   # upscaled like the benchmark's contents
            ups = []
            for x in pair:
                f, up = ctx.frame_from(x), ctx.create_frame(2 * w, 2 * h)
                ctx.scale(f, up)
                ctx.sync()
                ctx.destroy_frame(f)
                ups.append(up)
            frames.append(ups)
        mvs = [ctx.create_frame(2 * w, 2 * h, capi.FORMAT_MV_S8X2) for _ in range(2)]
        for j in range(2):                                    # workspaces, tables, verdicts: not in the measurement
            ctx.lane_select(j)
            ctx.motion(frames[j][0], frames[j][1], mvs[j], 8, 16.0)
        ctx.sync()
        for _ in range(rounds):
            ctx.lane_select(0)
            ctx.motion(frames[0][0], frames[0][1], mvs[0], 8, 16.0)
            ctx.comm_probe(8, 50)
            ctx.lane_select(1)
            ctx.motion(frames[1][0], frames[1][1], mvs[1], 8, 16.0)
            ms = ctx.comm_probe_ms()
            t1 = time.perf_counter()
            ctx.lane_sync()
            out.append((ms, (time.perf_counter() - t1) * 1e3))
        ctx.lane_select(0)
        ctx.sync()
        for x in [f for pair in frames for f in pair] + mvs:
            ctx.destroy_frame(x)
        ctx.lanes(1)
    return reserved, out


def test_a_broadcast_finds_a_cu_while_a_full_persistent_grid_runs(monkeypatch):
    """RCCL's device kernel cannot share a CU with a workgroup of the persistent prefilter kernel, and
    those stay for the whole launch -- so while a communicator exists the library's streams leave 8 CUs alone (CU masks).
    A communicator of one rank launches nothing for a broadcast; lfg_comm_probe puts a kernel of the same footprint where
    the broadcast would run.  With the reservation it runs as soon as it is ready, milliseconds before the call that holds
    the chip by then ends; without (LFG_COMM_CUS=0) it waits for room in that call's persistent launch -- printed only."""
    from linux_fg_amd import capi
    w, h = 1920, 1080
    prev = synth.make_prev(w, h, seed=811)
    noise = synth.noise_bytes(w, h, 861) % 17
    inputs = [(prev, np.clip(synth.translate(prev, (3, -2), 811).astype(np.int16) + noise.astype(np.int16) - 8, 0, 255).astype(np.uint8)),
              (prev, synth.noise_bytes(w, h, 862))]
    monkeypatch.delenv("LFG_COMM_CUS", raising=False)
    reserved, with_cus = _probe_behind_a_full_grid(capi, inputs)
    assert reserved == 8
    monkeypatch.setenv("LFG_COMM_CUS", "0")
    none, without = _probe_behind_a_full_grid(capi, inputs)
    assert none == 0
    print("probe ready -> done (device ms) / lane 1 behind the probe (host ms): 8 CUs kept", [(round(a, 3), round(b, 3)) for a, b in with_cus],
          "none kept", [(round(a, 3), round(b, 3)) for a, b in without])
    for took, ahead in with_cus:
        assert took < 0.15, with_cus                         # the probe's own 50 us and a launch (measured: 0.054 - 0.055)
        assert ahead > 0.5, with_cus                         # ... while the other call still has a millisecond or more to go
    # (without the reservation, measured: 0.055, 0.505, 0.482 ms -- the probe gets in when workgroups of the persistent launch happen to
    #  have left a CU of the shader engine its workgroups are handed to; nothing to assert on)


def test_the_cu_mask_of_a_communicator_and_its_return():
    from linux_fg_amd import capi
    with capi.Context(0) as ctx:
        full = ctx.comm_cu_mask()
        assert ctx.comm_reserved_cus() == 0 and full == [0xFFFFFFFF] * 8
        ctx.comm_init(1, 0, capi.Context.comm_unique_id())
        assert ctx.comm_reserved_cus() == 8 and ctx.comm_cu_mask() == [0xFFFFFF00] + [0xFFFFFFFF] * 7
        with pytest.raises(capi.LfgError):
            ctx.comm_cu_mask(words=4)                         # 256 CUs do not fit 128 bits
        a = synth.make_prev(320, 180, seed=9)
        f, up = ctx.frame_from(a), ctx.create_frame(640, 360)
        ctx.scale(f, up)                                      # kernels run on the masked stream as on any other
        first = ctx.download(up).copy()
        ctx.comm_destroy()
        assert ctx.comm_reserved_cus() == 0 and ctx.comm_cu_mask() == full
        ctx.scale(f, up)
        assert (ctx.download(up) == first).all()


def test_sixteen_cus_for_the_communicator(monkeypatch):
    """LFG_COMM_CUS=16: two CUs in each XCD, a probe of sixteen workgroups of RCCL's footprint runs at once while the lanes hold the chip."""
    from linux_fg_amd import capi
    monkeypatch.setenv("LFG_COMM_CUS", "16")
    w, h = 1920, 1080
    prev = synth.make_prev(w, h, seed=911)
    pair = (prev, synth.noise_bytes(w, h, 912))
    with capi.Context(0) as ctx:
        ctx.lanes(2)
        ctx.comm_init(1, 0, capi.Context.comm_unique_id())
        assert ctx.comm_reserved_cus() == 16 and ctx.comm_cu_mask()[0] == 0xFFFF0000
        ups = []
        for x in pair:
            f, up = ctx.frame_from(x), ctx.create_frame(2 * w, 2 * h)
            ctx.scale(f, up); ctx.sync(); ctx.destroy_frame(f); ups.append(up)
        mvs = [ctx.create_frame(2 * w, 2 * h, capi.FORMAT_MV_S8X2) for _ in range(2)]
        for j in range(2):
            ctx.lane_select(j)
            ctx.motion(ups[0], ups[1], mvs[j], 8, 16.0)
        ctx.sync()
        assert ctx.motion_plan()[1] == 480                    # two persistent workgroups on each of the other 240 CUs
        ctx.lane_select(0)
        ctx.motion(ups[0], ups[1], mvs[0], 8, 16.0)
        ctx.comm_probe(16, 50)
        ctx.lane_select(1)
        ctx.motion(ups[0], ups[1], mvs[1], 8, 16.0)               # 7 ms of a full persistent grid behind lane 0's
        assert ctx.comm_probe_ms() < 0.15
        ctx.lane_select(0)
        ctx.sync()
        for f in ups + mvs:
            ctx.destroy_frame(f)
        ctx.lanes(1)
