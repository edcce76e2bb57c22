"""Multi-GPU sharding of the hot path: one process per GPU, frame pairs are independent.

The path has exactly one exchange: a batch of frame pairs can share its previous frame, which then
travels from the rank that owns it to every other rank as ONE broadcast per step (RCCL over xGMI on
the GPU node: backend "nccl"; "gloo" on CPU for the tests).  There is no other collective on the data
path; the timing all-reduce in bench.py is measurement, not data.

``SharedFrameBroadcaster`` double-buffers that broadcast: while step k runs its kernels on slot
k % 2, the broadcast for step k + 1 is already in flight into the other slot (issued asynchronously
one step ahead), so on the GPU it overlaps with the step's compute.  torch.distributed orders the
collective after everything already enqueued on the caller's stream, which is what makes re-using a
slot two steps later safe (the kernels that read it were enqueued before the next broadcast into it).
"""
from __future__ import annotations

from typing import Callable, List, Optional


def stream_of_rank(rank: int, world_size: int, streams: int | None = None) -> List[int]:
    """Streams (independent frame-pair sequences) owned by ``rank``: stream i -> rank i % world_size.
    With the default ``streams = world_size`` that is one stream per GPU (BASELINE config 4)."""
    n = world_size if streams is None else streams
    return [s for s in range(n) if s % world_size == rank]


class SharedFrameBroadcaster:
    """Double-buffered broadcast of the shared previous frame from ``src`` to all ranks."""

    def __init__(self, slots, src: int = 0, dist=None, world_size: int = 1,
                 refill: Optional[Callable[[int, int], None]] = None):
        """``slots``: two tensors of equal shape (one when world_size == 1).  ``refill(step, slot)`` is
        called on the source rank before the broadcast for ``step`` is issued, to place that step's
        frame into ``slots[slot]`` (optional: the benchmark keeps a constant frame)."""
        self.slots = list(slots)
        self.src = src
        self.dist = dist
        self.world = world_size
        self.refill = refill
        self.pending = [None] * len(self.slots)
        self.issued_for = [None] * len(self.slots)
        if self.world > 1 and len(self.slots) != 2:
            raise ValueError("two slots are needed to overlap the broadcast with compute")

    def _issue(self, step: int):
        slot = step % len(self.slots)
        if self.world > 1:
            if self.refill is not None and self.dist.get_rank() == self.src:
                self.refill(step, slot)
            self.pending[slot] = self.dist.broadcast(self.slots[slot], src=self.src, async_op=True)
        elif self.refill is not None:
            self.refill(step, slot)
        self.issued_for[slot] = step

    def start(self, first_step: int = 0):
        self._issue(first_step)

    def acquire(self, step: int):
        """Returns the tensor holding the shared frame for ``step`` (waiting for its broadcast) and
        issues the broadcast for ``step + 1`` into the other slot."""
        slot = step % len(self.slots)
        if self.issued_for[slot] != step:
            self._issue(step)                         # not primed (first call without start())
        if self.pending[slot] is not None:
            self.pending[slot].wait()                 # orders the caller's stream after the broadcast
            self.pending[slot] = None
        if self.world > 1:
            self._issue(step + 1)
        return self.slots[slot]

    def drain(self):
        for i, p in enumerate(self.pending):
            if p is not None:
                p.wait()
                self.pending[i] = None
