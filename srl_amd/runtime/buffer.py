"""Sample batching in front of the trainer: ``batch_size`` per-agent ``[Tb, ...]`` samples -> one ``[Tb, B, ...]`` batch.

Counterpart of the reference's ``PriorityQueueBuffer`` (``base/buffer.py:62-172``): ``put`` stacks samples on
axis 1 (``:120-121``), entries are ordered by (reuses_left, receive_time) and served newest / least-used first
(``:24-31,142-162``), overflow drops the oldest entry (``:164-165``), an entry is served ``reuses`` times.
This is the row "sample ingest" of SURVEY.md 8(f)-1 in its simplest host form; the layout contract it
guarantees (time-major, batch on axis 1, every leaf contiguous) is what the trainer's kernels rely on.
"""
import bisect
import dataclasses
import time

import numpy as np

from srl_amd.namedarray import recursive_aggregate


@dataclasses.dataclass
class ReplayEntry:
    reuses_left: int
    receive_time: float
    sample: object
    reuses: int = 0

    def __lt__(self, other):
        return (self.reuses_left, self.receive_time) < (other.reuses_left, other.receive_time)


class PriorityQueueBuffer:

    def __init__(self, max_size=16, reuses=1, batch_size=None):
        self.max_size, self.reuses, self.batch_size = max_size, reuses, batch_size
        self._entries = []
        self._pending = []

    def qsize(self):
        return len(self._entries)

    def empty(self):
        return not self._entries

    def full(self):
        return len(self._entries) == self.max_size

    def put(self, x) -> bool:
        """Returns True when a full batch was formed."""
        if not self.batch_size:
            self._insert(ReplayEntry(self.reuses, time.time(), x))
            return False
        x.trainer_worker_recv_timestamp = np.full(shape=x.on_reset.shape, fill_value=int(time.time()), dtype=np.int64)
        self._pending.append(x)
        if len(self._pending) < self.batch_size:
            return False
        group, self._pending = self._pending[:self.batch_size], self._pending[self.batch_size:]
        batch = recursive_aggregate(group, lambda xs: np.stack(xs, axis=1))
        self._insert(ReplayEntry(self.reuses, time.time(), batch))
        return True

    def _insert(self, entry):
        bisect.insort(self._entries, entry)
        while len(self._entries) > self.max_size:
            self._entries.pop(0)

    def get(self) -> ReplayEntry:
        assert not self.empty(), "attempting to get from empty buffer."
        entry = self._entries.pop(-1)
        entry.reuses_left -= 1
        entry.reuses += 1
        if not self.full() and entry.reuses_left > 0:
            self._insert(entry)
        return entry
