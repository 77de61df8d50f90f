"""Pinned host staging for small tables that travel to the GPU with asynchronous copies.

An asynchronous host-to-device copy reads the pinned buffer when the STREAM reaches it, not when it is queued; the host of a training
loop runs ahead of the GPU (run_train waits for the forward pass only), so a single buffer rewritten every step is a race: the copy of
step i may pick up the values of step i+1.  `PinnedRing` hands out slots in turn and fences each one with an event recorded after its
copy: a slot is rewritten only once the copy that read it has completed (the wait is normally over long before it is asked for).
"""
import torch


class PinnedRing:
    def __init__(self, shape, dtype, slots=4):
        self.slots = [torch.zeros(shape, dtype=dtype).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.pos = 0

    def acquire(self, min_rows=None):
        """-> (index, pinned tensor) of the next slot, safe to overwrite.  min_rows: grow the slot's first dimension if it is smaller."""
        i = self.pos % len(self.slots)
        self.pos += 1
        if self.events[i] is not None:
            self.events[i].synchronize()
        if min_rows is not None and self.slots[i].shape[0] < min_rows:
            self.slots[i] = torch.zeros((min_rows,) + tuple(self.slots[i].shape[1:]), dtype=self.slots[i].dtype).pin_memory()
        return i, self.slots[i]

    def sent(self, i, stream=None):
        """Call right after queueing the copy that reads slot i (on the current stream unless given)."""
        if self.events[i] is None:
            self.events[i] = torch.cuda.Event()
        self.events[i].record(stream) if stream is not None else self.events[i].record()
