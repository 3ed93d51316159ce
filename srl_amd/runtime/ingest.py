"""Sample ingest ring: actor trajectories -> ``[Tb, B, ...]`` SoA in pinned host memory -> HBM, without the
axis-1 ``np.stack`` of the reference's buffer.

The reference stacks ``batch_size`` per-agent samples with ``np.stack(xs, axis=1)`` when a batch is complete
(``base/buffer.py:120-121``): at Atari sizes that is a second pass over 1.86 GB of frames per batch, into
pageable memory, followed by a pageable H2D copy and a widening of every leaf to float32
(``api/trainer.py:211-228``).  Its shared-memory dock already has the right shape for something better -- one
block per flattened key, ``[T, qsize, ...]`` (``base/shared_memory.py:55-83``).  This ring is that shape in
pinned memory:

* one pinned block per flattened key and slot, ``[Tb, B, ...]`` in the leaf's wire dtype (uint8 frames and
  flags stay uint8);
* ``put_column`` writes one trajectory ``[Tb, ...]`` straight into column ``b`` of the slot being filled
  (a strided copy: the only host pass over the data); ``put_wire`` does the same from the ``raw_bytes`` wire
  format (``namedarray.dumps(..., "raw_bytes")``; reference ``base/namedarray.py:115-139,178-191``) without
  materialising an intermediate namedarray;
* a full slot is handed to the trainer either as zero-copy host views (``get``) or as device leaves
  (``get_device``): one asynchronous copy per leaf on a side stream into per-slot device buffers, so the copy of
  batch k+1 overlaps the update on batch k.  ``MultiAgentPPO.step`` takes device-resident samples as they are.

In-process and thread-safe (actors in threads, or a receiver thread decoding the sample stream); the blocks can
be exported to actor processes with ``multiprocessing.shared_memory`` by the caller (``block_specs``).
"""
import threading
from collections import OrderedDict, deque
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from srl_amd import namedarray as na


def _flat_leaves(x) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((k, v) for k, v in na.flatten(x) if v is not None)


class SampleRing:

    def __init__(self, template, batch_size: int, slots: int = 2, device: Optional[str] = None, pin: Optional[bool] = None,
                 obs_ring=None):
        """``template``: one trajectory (``SampleBatch`` / ``NamedArray`` with leaves ``[Tb, ...]``) fixing keys, dtypes
        and trailing shapes.  ``device``: where ``get_device`` puts batches (None: host only).  ``obs_ring``: the policy's
        HBM observation ring (``runtime/obs_ring.py``): ``get_device`` then binds a batch's observations to the rows the
        rollout left there (by the ``analyzed_result.obs_ref`` stamps the samples carry) and uploads only what is not --
        the scalar leaves, and any row whose stamp is no longer alive."""
        if slots < 1 or batch_size < 1:
            raise ValueError("slots and batch_size must be positive")
        self._template = template
        self.batch_size, self.slots, self.device = int(batch_size), int(slots), device
        leaves = _flat_leaves(template)
        if not leaves:
            raise ValueError("template has no array leaves")
        self.Tb = int(next(iter(leaves.values())).shape[0])
        pin = (device is not None and torch.cuda.is_available()) if pin is None else pin
        self._host: List["OrderedDict[str, torch.Tensor]"] = []
        for _ in range(self.slots):
            blk = OrderedDict()
            for k, v in leaves.items():
                v = np.asarray(v)
                if v.shape[0] != self.Tb:
                    raise ValueError(f"leaf `{k}` has {v.shape[0]} rows, expected {self.Tb}")
                t = torch.empty((self.Tb, self.batch_size, *v.shape[1:]), dtype=torch.from_numpy(v[:0]).dtype)
                blk[k] = t.pin_memory() if pin else t
            self._host.append(blk)
        self._np = [OrderedDict((k, t.numpy()) for k, t in blk.items()) for blk in self._host]
        self._dev: List[Optional["OrderedDict[str, torch.Tensor]"]] = [None] * self.slots
        self._copied: List[Optional[torch.cuda.Event]] = [None] * self.slots
        self.obs_ring = obs_ring
        self._ring_keys = frozenset(f"obs.{k}" for k in obs_ring.keys()) & frozenset(leaves) if obs_ring is not None else frozenset()
        self._leases = [None] * self.slots
        self._fallback_copied = [None] * self.slots  # event after the plain copies of a slot the observation ring could not serve
        self._stream = None
        self._lock = threading.Lock()
        self._free = deque(range(self.slots))
        self._full = deque()
        self._filling: Optional[int] = None
        self._count = 0
        # columns whose write has FINISHED, per slot: a slot is published only when all of them have (a slot whose
        # last column was merely claimed may still have other threads writing theirs)
        self._written = [0] * self.slots

    # ------------------------------------------------------------------ introspection
    def block_specs(self) -> Dict[str, Tuple[Tuple[int, ...], str]]:
        """key -> (shape, dtype) of one slot's blocks (what an exporter to actor processes needs)."""
        return {k: (tuple(v.shape), str(v.dtype)) for k, v in self._np[0].items()}

    def nbytes(self) -> int:
        return sum(v.nbytes for blk in self._np for v in blk.values())

    def host_blocks(self, slot: int) -> "OrderedDict[str, np.ndarray]":
        """The slot's pinned blocks by flattened key, ``[Tb, B, ...]`` numpy views (what an exporter to actor processes
        maps; benchmarks write the stamps of a synthetic rollout here -- followed by ``invalidate_copies``: a complete
        slot's copy may have been started ahead of time)."""
        return self._np[slot]

    def host_tensors(self, slot: int) -> "OrderedDict[str, torch.Tensor]":
        """The same blocks as pinned torch tensors (a copy from them is an asynchronous DMA; from their numpy views torch
        stages through a pageable buffer)."""
        return self._host[slot]

    def invalidate_copies(self):
        """Forget every host-to-device copy already started for a slot that is not checked out: its host blocks were
        modified afterwards (``host_blocks``), or the set of leaves to copy changed.  The next ``get_device`` copies again."""
        for slot in range(self.slots):
            ev = self._copied[slot]
            if ev is not None:
                ev.synchronize()
                self._copied[slot] = None
            self._dev[slot] = None

    def attach_obs_ring(self, obs_ring):
        """Serve the observation leaves from ``obs_ring`` from now on (None: copy them like every other leaf).  Call it
        with no batch checked out."""
        self.obs_ring = obs_ring
        keys = frozenset(f"obs.{k}" for k in obs_ring.keys()) & frozenset(self._np[0]) if obs_ring is not None else frozenset()
        self._ring_keys = keys
        self.invalidate_copies()  # copies started ahead of time carry the old key set (and possibly old stamps)

    def ready(self) -> int:
        with self._lock:
            return len(self._full)

    # ------------------------------------------------------------------ producer side
    def _commit(self, slot: int, columns: int = 1) -> Optional[int]:
        """Called after a column's write has finished; the writer of the last column to finish publishes the slot."""
        with self._lock:
            self._written[slot] += columns
            if self._written[slot] == self.batch_size:
                self._written[slot] = 0
                self._full.append(slot)
                if slot == self._filling:
                    self._filling = None
                return slot
        return None

    def _claim(self) -> Tuple[int, int]:
        """(slot, column) for the next trajectory; raises when every slot is full and unreleased.  Once every column
        of the slot being filled is claimed the next claim opens a new slot, whether or not the writers are done."""
        with self._lock:
            if self._filling is not None and self._count == self.batch_size:
                self._filling = None
            if self._filling is None:
                if not self._free:
                    raise BufferError("sample ring is full: release a slot (trainer is behind the actors)")
                self._filling = self._free.popleft()
                self._count = 0
            slot, col = self._filling, self._count
            self._count += 1
            return slot, col

    def _check_column(self, leaves):
        """Every leaf of a trajectory must fit its column BEFORE one is claimed: a write that fails after the claim would
        leave the column claimed and never committed, and the slot lost to the ring for good."""
        blk = self._np[0]
        for k, arr in leaves:
            want = blk[k].shape[:1] + blk[k].shape[2:]
            if tuple(np.shape(arr)) != want:
                raise ValueError(f"leaf `{k}` has shape {tuple(np.shape(arr))}, the ring's columns take {want}")

    def put_column(self, traj) -> Optional[int]:
        """Write one trajectory (leaves ``[Tb, ...]``) into the next column; returns the slot id when that
        completed a batch, else None."""
        leaves = _flat_leaves(traj)
        if set(leaves) != set(self._np[0]):  # before a column is claimed: a bad trajectory must not leave a hole
            raise KeyError(f"trajectory keys differ from the ring's: {sorted(set(leaves) ^ set(self._np[0]))}")
        self._check_column(leaves.items())
        slot, col = self._claim()
        blk = self._np[slot]
        for k, dst in blk.items():
            dst[:, col] = leaves[k]  # strided write; numpy converts dtypes if the producer's differ
        return self._commit(slot)

    def put_wire(self, chunks: List[bytes]) -> Optional[int]:
        """Same as ``put_column`` for one trajectory in the ``raw_bytes`` wire format: every leaf is decoded with
        ``np.frombuffer`` (no copy) and written straight into its column."""
        if not na.is_raw_bytes(chunks):
            return self.put_column(na.loads(chunks))
        leaves = [(k, arr) for k, arr in na.iter_raw_leaves(chunks) if arr is not None]  # zero-copy views
        keys = {k for k, _ in leaves}
        if keys != set(self._np[0]):  # before a column is claimed: a bad message must not leave a hole in the slot
            raise KeyError(f"wire trajectory keys differ from the ring's: {sorted(keys ^ set(self._np[0]))}")
        self._check_column(leaves)
        slot, col = self._claim()
        blk = self._np[slot]
        for k, arr in leaves:
            blk[k][:, col] = arr
        return self._commit(slot)

    def put_batch(self, batch) -> int:
        """A whole ``[Tb, B, ...]`` batch that arrives at once (e.g. from a sample stream) into the next free slot."""
        leaves = _flat_leaves(batch)
        if set(leaves) != set(self._np[0]):
            raise KeyError(f"batch keys differ from the ring's: {sorted(set(leaves) ^ set(self._np[0]))}")
        with self._lock:
            if self._filling is not None and 0 < self._count < self.batch_size:
                raise BufferError("a slot is being filled column by column")
            if self._filling is not None and self._count == self.batch_size:
                self._filling = None
            if self._filling is None:
                if not self._free:
                    raise BufferError("sample ring is full: release a slot (trainer is behind the actors)")
                self._filling = self._free.popleft()
            slot, self._count = self._filling, self.batch_size
        for k, dst in self._np[slot].items():
            if isinstance(leaves[k], torch.Tensor):  # device-resident batch: D2H straight into the pinned block
                self._host[slot][k].copy_(leaves[k])
            else:
                dst[...] = leaves[k]
        return self._commit(slot, self.batch_size)

    def recycle(self, slot: int):
        """Benchmarks: make a released slot complete again with the data it still holds (no host pass)."""
        with self._lock:
            self._free.remove(slot)
            self._full.append(slot)

    # ------------------------------------------------------------------ consumer side
    def _take(self) -> int:
        with self._lock:
            if not self._full:
                raise LookupError("no complete batch in the ring")
            return self._full.popleft()

    def _wrap(self, slot: int, leaves):
        """The template's structure (same NamedArray subclasses, None entries kept) over the slot's blocks."""

        def rebuild(node, prefix):
            if isinstance(node, na.NamedArray):
                fields = {k: rebuild(v, f"{prefix}{k}.") for k, v in node.items()}
                try:
                    return type(node)(**fields)
                except TypeError:
                    return na.NamedArray(**fields)
            return None if node is None else leaves[prefix[:-1]]

        batch = rebuild(self._template, "")
        batch.register_metadata(ring_slot=slot)
        return batch

    def get(self):
        """Oldest complete batch as zero-copy host views ``[Tb, B, ...]`` (numpy).  ``release`` it when done."""
        slot = self._take()
        return self._wrap(slot, self._np[slot])

    def _start_copy(self, slot: int):
        """Enqueue the H2D copies of a complete slot on the side stream (idempotent until the slot is released).  Leaves
        that an observation ring serves are not copied here (``_bind_or_copy``)."""
        if self._copied[slot] is not None:
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=self.device)
        if self._dev[slot] is None:
            self._dev[slot] = OrderedDict((k, torch.empty(t.shape, dtype=t.dtype, device=self.device))
                                          for k, t in self._host[slot].items() if k not in self._ring_keys)
        with torch.cuda.stream(self._stream):
            for k, t in self._host[slot].items():
                if k not in self._ring_keys:
                    self._dev[slot][k].copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._copied[slot] = ev

    def _bind_or_copy(self, slot: int):
        """Observation leaves of a taken slot: ``RingObs`` over the rows the rollout left in HBM (rows without a live
        stamp are uploaded from the slot's host block on the way), or -- when the ring cannot serve the batch -- plain
        device copies on the caller's stream."""
        refs = self._np[slot].get("analyzed_result.obs_ref")
        host_obs = {k[4:]: self._host[slot][k] for k in self._ring_keys}
        refs_dev = self._dev[slot].get("analyzed_result.obs_ref") if self._dev[slot] is not None else None
        bound = self.obs_ring.bind(refs, host_obs, refs_device=refs_dev)
        if bound is None:
            # plain copies on the caller's stream, from this slot's pinned host block: `release` must not hand the block
            # back to the producers before they (and whatever patch uploads a bind that failed midway had enqueued) are done
            out = {k: self._host[slot][k].to(self.device, non_blocking=True) for k in self._ring_keys}
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._fallback_copied[slot] = ev
            return out
        rows, lease = bound
        self._leases[slot] = lease
        return {f"obs.{k}": v for k, v in rows.items()}

    def get_device(self):
        """Oldest complete batch with device leaves.  Its copies run on the ring's side stream and the caller's
        current stream waits on them (no host synchronisation); the copies of the NEXT complete batch, if there is
        one, are started right away, so they overlap the update the caller is about to enqueue on this batch.
        ``release`` the slot once the batch has been consumed."""
        if self.device is None or not torch.cuda.is_available():
            raise RuntimeError("get_device needs a ring constructed with a GPU `device`")
        slot = self._take()
        self._start_copy(slot)
        torch.cuda.current_stream(self.device).wait_event(self._copied[slot])
        with self._lock:
            nxt = self._full[0] if self._full else None
        if nxt is not None:
            self._start_copy(nxt)
        leaves = self._dev[slot]
        if self._ring_keys:
            leaves = dict(leaves)
            leaves.update(self._bind_or_copy(slot))
        return self._wrap(slot, leaves)

    def release(self, batch_or_slot):
        """Return a slot to the producers.  Host blocks are reusable once the H2D copy has finished (waited here);
        the device buffers of the slot are reused by the next ``get_device`` of the same slot, which is ordered
        after the consumer's work by the caller releasing only after it has enqueued that work."""
        slot = batch_or_slot if isinstance(batch_or_slot, int) else batch_or_slot.metadata["ring_slot"]
        ev = self._copied[slot]
        if ev is not None:
            ev.synchronize()
            self._copied[slot] = None
        ev = self._fallback_copied[slot]
        if ev is not None:
            ev.synchronize()
            self._fallback_copied[slot] = None
        if self._leases[slot] is not None:  # the observation rows may be lapped once the consumer's work has run
            lease, self._leases[slot] = self._leases[slot], None
            if lease.uploaded is not None:
                lease.uploaded.synchronize()  # patch uploads read this slot's host block
            self.obs_ring.release(lease)
        if self._dev[slot] is not None and self._stream is not None:
            # the next copy into these device buffers must not overtake kernels still reading them
            self._stream.wait_stream(torch.cuda.current_stream(self.device))
        with self._lock:
            if slot in self._free or slot in self._full or slot == self._filling:
                raise ValueError(f"slot {slot} is not checked out")
            self._free.append(slot)
