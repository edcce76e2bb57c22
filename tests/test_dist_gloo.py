"""The N > 1 path on CPU: two processes over gloo run the same double-buffered shared-frame
broadcast and per-rank sharding that bench.py uses on the GPU node (there the transport is the C-ABI's
lfg_broadcast_frame = RCCL; here torch.distributed over gloo carries the bytes).  Checks that every rank sees the
source rank's frame for every step, that slots are never clobbered before use, the communicator-id exchange
bench.py does before lfg_comm_init, the C-ABI's argument validation without a GPU, and the max-over-ranks /
whole-job aggregation bench.py reports."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from linux_fg_amd import sharding, synth

STEPS = 5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h = 32, 18
        frames = [torch.from_numpy(synth.make_prev(w, h, synth.BASE_SEED + 100 + k)) for k in range(STEPS + 1)]
        slots = [torch.zeros((h, w, 4), dtype=torch.uint8) for _ in range(2)]

        def refill(step, slot):                      # only ever called on the source rank
            slots[slot].copy_(frames[step])

        # the id exchange of bench.py: rank 0 makes 128 bytes, everyone ends up with the same ones.  (A real id needs a
        # GPU -- RCCL initialises HIP -- so here rank 0 makes recognisable bytes instead.)
        comm_id = sharding.exchange_comm_id(dist, lambda: bytes((7 * i + 3) & 0xFF for i in range(128)), src=0)
        assert comm_id == bytes((7 * i + 3) & 0xFF for i in range(128))
        with pytest.raises(ValueError):
            if rank == 0:
                sharding.exchange_comm_id(dist, lambda: b"short", src=0)
            else:
                raise ValueError("(only the source rank validates the id it made)")
        # the C-ABI's communicator calls refuse bad arguments before they touch RCCL or a GPU
        from linux_fg_amd import capi
        lib = capi.load()
        import ctypes
        idbuf = ctypes.create_string_buffer(comm_id, 128)
        assert lib.lfg_comm_init(None, world, rank, idbuf) == -1          # no context
        assert lib.lfg_comm_unique_id(None) == -1
        assert lib.lfg_comm_ranks(None) == 0 and lib.lfg_comm_rank(None) == -1
        assert lib.lfg_comm_wait(None) == -1 and lib.lfg_broadcast_frame(None, None, 0) == -1

        transport = sharding.TorchTransport(dist, slots, src=0)
        b = sharding.SharedFrameBroadcaster(2, transport, world_size=world, is_source=rank == 0, refill=refill)
        b.start(0)
        sums = []
        mine = sharding.stream_of_rank(rank, world)
        for k in range(STEPS):
            shared = slots[b.acquire(k)]
            assert torch.equal(shared, frames[k]), f"rank {rank} step {k}: wrong shared frame"
            # stand-in for the per-rank kernels: combine the shared frame with this rank's own stream
            own = torch.from_numpy(synth.make_prev(w, h, synth.BASE_SEED + mine[0]))
            sums.append(int((shared.to(torch.int64) + own.to(torch.int64)).sum()))
        b.drain()
        # bench.py's aggregation: elapsed = max over ranks, value = world * steps / elapsed
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, sums, float(t.item()), mine))
    finally:
        dist.destroy_process_group()


def test_two_rank_shared_frame_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert [r[3] for r in results] == [[0], [1]]                     # one stream per rank
    assert all(r[2] == 2.0 for r in results)                         # max over ranks reached everyone
    assert len(results[0][1]) == STEPS and results[0][1] != results[1][1]   # different own streams
