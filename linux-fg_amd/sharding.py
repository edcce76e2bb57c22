"""Multi-GPU sharding of the hot path: one process per GPU, frame pairs are independent.

The path has exactly one exchange: a batch of frame pairs can share its previous frame, which then
travels from the rank that owns it to every other rank as ONE broadcast per step.  On the GPU node that
broadcast is the C-ABI's ``lfg_broadcast_frame`` (RCCL over xGMI on the context's communication stream,
include/linuxfg_hip.h) -- ``CapiTransport``; the CPU tests drive the same scheduling over gloo --
``TorchTransport``.  There is no other collective on the data path; the barrier and the max-over-ranks
of the elapsed time in bench.py are measurement, not data.

``SharedFrameBroadcaster`` double-buffers the broadcast: while step k runs its kernels on slot k % 2,
the broadcast for step k + 1 is already in flight into the other slot (issued one step ahead), so on
the GPU it overlaps with the step's compute.  A transport orders each broadcast after everything the
consumer has enqueued so far, which is what makes re-using a slot two steps later safe (the kernels
that read it were enqueued before the next broadcast into it).
"""
from __future__ import annotations

from typing import Callable, List, Optional


def stream_of_rank(rank: int, world_size: int, streams: int | None = None) -> List[int]:
    """Streams (independent frame-pair sequences) owned by ``rank``: stream i -> rank i % world_size.
    With the default ``streams = world_size`` that is one stream per GPU (BASELINE config 4)."""
    n = world_size if streams is None else streams
    return [s for s in range(n) if s % world_size == rank]


def exchange_comm_id(dist, make_id: Callable[[], bytes], src: int = 0, nbytes: int = 128) -> bytes:
    """Rank ``src`` makes the communicator id (``capi.Context.comm_unique_id``) and every rank receives the same
    ``nbytes`` bytes over an existing torch.distributed group (any backend with CPU tensors: gloo)."""
    import torch
    buf = torch.zeros(nbytes, dtype=torch.uint8)
    if dist.get_rank() == src:
        raw = make_id()
        if len(raw) != nbytes:
            raise ValueError(f"communicator id must be {nbytes} bytes, got {len(raw)}")
        buf = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone()
    dist.broadcast(buf, src=src)
    return bytes(buf.numpy().tobytes())


class CapiTransport:
    """The product transport: lfg_broadcast_frame / lfg_comm_wait on a context that has a communicator."""

    def __init__(self, ctx, frames, src: int = 0, behind_selected_lane_only: bool = False):
        """``behind_selected_lane_only``: lfg_broadcast_frame_lane -- for a step that has already ordered the selected lane
        behind the last readers of the slot about to be overwritten (``lane_wait`` for the previous step's upscales, which
        read slot (k + 1) % 2 last: SharedFrameBroadcaster issues the broadcast for step k + 1 inside step k's ``acquire``).
        The default orders a broadcast behind everything every lane has been given: always safe, but with three frames in
        flight step k + 1 then waits for all of step k - 1."""
        self.ctx, self.frames, self.src, self.lane_only = ctx, list(frames), src, behind_selected_lane_only

    def issue(self, slot: int):
        if self.lane_only:
            self.ctx.broadcast_frame_lane(self.frames[slot], self.src)
        else:
            self.ctx.broadcast_frame(self.frames[slot], self.src)

    def wait(self, slot: int):
        self.ctx.comm_wait()


class TorchTransport:
    """torch.distributed broadcast of tensors (gloo in the CPU tests)."""

    def __init__(self, dist, tensors, src: int = 0):
        self.dist, self.tensors, self.src = dist, list(tensors), src
        self.pending = [None] * len(self.tensors)

    def issue(self, slot: int):
        self.pending[slot] = self.dist.broadcast(self.tensors[slot], src=self.src, async_op=True)

    def wait(self, slot: int):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None


class SharedFrameBroadcaster:
    """Double-buffered broadcast of the shared previous frame from the source rank to all ranks."""

    def __init__(self, n_slots: int, transport=None, world_size: int = 1, is_source: bool = True,
                 refill: Optional[Callable[[int, int], None]] = None):
        """``n_slots``: 2 (1 is enough when world_size == 1).  ``transport``: CapiTransport / TorchTransport (None
        when world_size == 1).  ``refill(step, slot)`` is called on the source rank before the broadcast for ``step``
        is issued, to place that step's frame into slot ``slot`` (optional: the benchmark keeps a constant frame)."""
        self.n = n_slots
        self.transport = transport
        self.world = world_size
        self.is_source = is_source
        self.refill = refill
        self.issued_for = [None] * n_slots
        self.in_flight = [False] * n_slots
        if self.world > 1 and (n_slots != 2 or transport is None):
            raise ValueError("two slots and a transport are needed to overlap the broadcast with compute")

    def _issue(self, step: int):
        slot = step % self.n
        if self.refill is not None and (self.world == 1 or self.is_source):
            self.refill(step, slot)
        if self.world > 1:
            self.transport.issue(slot)
            self.in_flight[slot] = True
        self.issued_for[slot] = step

    def start(self, first_step: int = 0):
        self._issue(first_step)

    def acquire(self, step: int) -> int:
        """Returns the slot holding the shared frame for ``step`` (its broadcast has been waited for, i.e. everything
        the consumer enqueues from now on comes after it) and issues the broadcast for ``step + 1`` into the other slot."""
        slot = step % self.n
        if self.issued_for[slot] != step:
            self._issue(step)                         # not primed (first call without start())
        if self.in_flight[slot]:
            self.transport.wait(slot)
            self.in_flight[slot] = False
        if self.world > 1:
            self._issue(step + 1)
        return slot

    def drain(self):
        for slot in range(self.n):
            if self.in_flight[slot]:
                self.transport.wait(slot)
                self.in_flight[slot] = False
