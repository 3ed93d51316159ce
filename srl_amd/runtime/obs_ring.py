"""HBM observation ring: the frames ``rollout`` uploaded stay on the device for the trainer.

In the reference every observation crosses the host-device link twice: once when the policy worker runs inference on
it (``actor_critic_policy.py:467-469``) and once more inside the training sample (``api/trainer.py:211-228``).  At Atari
sizes the second crossing is 14.9 GB per update -- 261 ms over a 57 GB/s link against 180 ms of compute -- and it stays
the bound at every GPU count, because copy and compute shrink together.  Nothing in the path needs it: the bytes the
trainer wants are the bytes the rollout already put in HBM.

``ObsRing`` keeps them.  It is a ring of fixed-size rows per observation key, allocated by a monotonically increasing
*sequence number* (slot = sequence mod capacity):

* ``put`` (called by ``ActorCriticPolicy.rollout`` on every inference batch) takes the batch's rows -- device tensors,
  typically the ``InferenceBatcher``'s staging block -- and writes them into the next run of slots **in the layout the
  network's first layer reads**: for a strided first convolution that is the space-to-depth re-tiling plus the
  whole-observation LayerNorm statistics (``srl_obs_space_to_depth``; the pass the inference forward needs anyway, now
  writing into the ring instead of a scratch buffer), raw rows otherwise.  It returns the rows' sequence numbers; the
  rollout hands them back to the actor as ``analyzed_result.obs_ref``, a leaf the actor stores verbatim in every step
  (``actor_worker.py:521-535`` copies the whole response into the memory step), so they arrive at the trainer inside
  the sample, ``[Tb, B, 1]`` int64, with no change on the actor side.
* ``bind`` (called by ``SampleRing.get_device``) checks a sample's references -- a row is alive iff no later allocation
  has lapped it -- uploads only the rows that are NOT alive (terminal observations never sent for inference, rows a
  slow sample lost to the ring, or everything when the sample carries no references: the full-copy path, through the
  same kernels) into a *patch area* behind the ring that belongs to this sample alone, and returns a ``RingObs`` per
  key: storage + an int32 slot index on the device.  The trainer's row
  chunks gather their staged rows from there (``srl_gather_rows``: the same bytes per row as the space-to-depth pass it
  replaces) and never see the host copy of the frames.
* a *lease* pins the rows of a bound sample until ``release``: an allocation that would lap leased rows raises
  ``BufferError`` instead of corrupting a step in flight.

Sharding: one ring per GPU, holding the env columns that GPU both serves inference for and trains on.
In-process (inference thread + trainer thread, as in the reference's local mode); thread-safe.
"""
import itertools
import threading
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from srl_amd import hip


class RingObs:
    """Observation rows that live in an ``ObsRing``: ``[*lead, *raw_shape]`` in the eyes of the sample, an int32 slot per
    row (``index``, device) or one contiguous run of slots (``span``) underneath.  Supports what the trainer does to an
    observation leaf: slicing the leading dimension, folding / flattening the leading dimensions, concatenation."""

    def __init__(self, ring: "ObsRing", key: str, lead: Tuple[int, ...], index: Optional[torch.Tensor] = None,
                 span: Optional[int] = None):
        self.ring, self.key, self.lead = ring, key, tuple(int(d) for d in lead)
        self.index, self.span = index, span
        assert (index is None) != (span is None)

    # ---- the tensor-like surface the trainer uses
    @property
    def shape(self):
        return (*self.lead, *self.ring.raw_shape[self.key])

    @property
    def dtype(self):
        return self.ring.dtype[self.key]

    @property
    def device(self):
        return torch.device(self.ring.device)

    is_cuda = True

    @property
    def rows(self) -> int:
        return int(np.prod(self.lead, dtype=np.int64))

    @property
    def layout(self):
        return self.ring.layout[self.key]

    def _flat_index(self) -> torch.Tensor:
        if self.index is None:  # a run of slots: materialised only if somebody slices it unevenly
            return torch.arange(self.span, self.span + self.rows, dtype=torch.int32, device=self.ring.device)
        return self.index.reshape(-1)

    def __getitem__(self, loc):
        if not isinstance(loc, slice):
            raise TypeError("RingObs supports slices of its leading dimension only")
        start, stop, step = loc.indices(self.lead[0])
        if step != 1:
            raise TypeError("RingObs slices must be contiguous")
        n0 = max(stop - start, 0)
        inner = int(np.prod(self.lead[1:], dtype=np.int64))
        if self.span is not None:
            return RingObs(self.ring, self.key, (n0, *self.lead[1:]), span=self.span + start * inner)
        return RingObs(self.ring, self.key, (n0, *self.lead[1:]), index=self.index.reshape(self.lead[0], -1)[start:stop])

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        raw = tuple(self.ring.raw_shape[self.key])
        if len(shape) <= len(raw) or tuple(shape[len(shape) - len(raw):]) != raw:
            raise ValueError(f"RingObs.reshape keeps the row shape {raw}; got {shape}")
        lead = shape[:len(shape) - len(raw)]
        if -1 in lead:
            known = int(np.prod([d for d in lead if d != -1], dtype=np.int64))
            lead = tuple(self.rows // max(known, 1) if d == -1 else d for d in lead)
        if int(np.prod(lead, dtype=np.int64)) != self.rows:
            raise ValueError(f"cannot reshape {self.rows} rows into {lead}")
        if self.span is not None:
            return RingObs(self.ring, self.key, lead, span=self.span)
        return RingObs(self.ring, self.key, lead, index=self.index.reshape(-1))

    @staticmethod
    def cat(parts):
        """Concatenation along the leading dimension (the burn-in windows of a recurrent policy)."""
        first = parts[0]
        idx = torch.cat([p._flat_index().reshape(p.lead[0], -1) for p in parts], dim=0)
        return RingObs(first.ring, first.key, (idx.shape[0], *first.lead[1:]), index=idx.contiguous())

    # ---- what the network reads
    def resolve(self, ws, name: str):
        """Contiguous staged rows ``[n, ...]`` (+ per-row LayerNorm statistics for the ``s2d`` layout, else None, None):
        views of the ring for a run of slots, gathered into the workspace buffers ``name + .s2d / .mean / .rstd`` otherwise."""
        r, k, n = self.ring, self.key, self.rows
        store = r.storage[k]
        row_elems = store[0].numel()
        stats = r.layout[k][0] == "s2d"
        if self.span is not None:
            s0 = self.span
            return (store[s0:s0 + n], r.mean[k][s0:s0 + n] if stats else None, r.rstd[k][s0:s0 + n] if stats else None)
        idx = self.index.reshape(-1)
        frames = ws.get(f"{name}.s2d", n * row_elems, dtype=store.dtype)[:n * row_elems].view(n, *store.shape[1:])
        hip.gather_rows(store.data_ptr(), row_elems * store.element_size(), idx, n, frames.data_ptr())
        if not stats:
            return frames, None, None
        mean = ws.get(f"{name}.mean", n)
        rstd = ws.get(f"{name}.rstd", n)
        hip.gather_rows(r.mean[k].data_ptr(), 4, idx, n, mean.data_ptr())
        hip.gather_rows(r.rstd[k].data_ptr(), 4, idx, n, rstd.data_ptr())
        return frames, mean, rstd

    def in_place(self):
        """``(storage rows, mean, rstd, int32 slot index [n])`` of the ``s2d`` layout for a kernel that addresses its rows
        through the index: nothing is copied."""
        r, k = self.ring, self.key
        return r.storage[k], r.mean[k], r.rstd[k], self._flat_index().contiguous()

    def gather_raw(self, ws, name: str) -> torch.Tensor:
        """The rows as one contiguous raw tensor ``[n, *raw_shape]`` (``raw`` layout only)."""
        if self.layout[0] != "raw":
            raise hip.HipError(f"observation `{self.key}` is staged in the `{self.layout[0]}` layout: no raw rows in the ring")
        frames, _, _ = self.resolve(ws, name)
        return frames.view(self.rows, *self.ring.raw_shape[self.key])


class ObsLease:
    """What ``bind`` pinned for one sample: ring rows from sequence ``min_seq`` on, and a run of the patch area."""

    def __init__(self, ring, min_seq):
        self.ring, self.min_seq = ring, min_seq
        self.patch_start = None  # first patch-area sequence number this lease holds (None: no patched rows)
        self.uploaded = None  # event after the uploads `bind` issued from the caller's host blocks (None: there were none)


class _Circular:
    """Sequence-number allocator of a circular region: slot = sequence % capacity, a run never straddles the end."""

    def __init__(self, capacity):
        self.capacity, self.head = int(capacity), 0

    def alloc(self, n, floor, what, check=None):
        """``check(seq)``: the caller's last word on the run [seq, seq + n) before the head moves (raises to refuse it: nothing
        has been consumed then)."""
        if n > self.capacity:
            raise BufferError(f"{n} rows do not fit {what} of {self.capacity} rows")
        seq = self.head
        if seq % self.capacity + n > self.capacity:  # the tail is skipped: its sequence numbers are consumed
            seq = (seq // self.capacity + 1) * self.capacity
        if floor is not None and seq + n - self.capacity > floor:
            raise BufferError(f"{what}: the allocation would overwrite rows a training step has leased "
                              f"(capacity {self.capacity} rows)")
        if check is not None:
            check(seq)
        self.head = seq + n
        return seq


_GENERATIONS = itertools.count(1)


class WholeStackNeeded(LookupError):
    """A stack-aware rollout request (newest planes + ``ring_prev`` stamps) could not be served: a predecessor is no longer in
    the ring (lapped, another ring's stamp, or the -1 an unstaged row got from ``put_or_skip`` -- such stamps must not be sent
    as ``ring_prev``), or the ring is full of leased rows.  Nothing was staged and no ring slot was consumed; the client sends
    the same request again with whole frame stacks (no ``ring_prev``)."""


class ObsRing:
    GEN_SHIFT = 44                      # sequence numbers below 2^44 rows (17 T): more than any run stages
    SEQ_MASK = (1 << GEN_SHIFT) - 1

    def __init__(self, layout: Dict[str, tuple], raw_shape: Dict[str, Tuple[int, ...]], capacity_rows: int, device: str,
                 patch_rows: Optional[int] = None):
        """``layout``: key -> ("s2d", block) | ("raw",) as ``HipNet.obs_stage_layout()`` reports it; ``raw_shape``: key -> the
        observation's shape as the environment produces it.  ``patch_rows``: size of the patch area behind the ring, where
        ``bind`` stages the rows of a sample that are not alive in the ring (a lapped row cannot be re-staged in the ring
        itself: the rows it would overwrite are the sample's own next-oldest ones); default capacity / 8.  Storage is
        allocated at the first ``put`` of a key (its dtype -- uint8 frames or float32 -- is the producer's)."""
        if capacity_rows < 1:
            raise ValueError("capacity_rows must be positive")
        self.layout = {k: tuple(v) for k, v in layout.items()}
        self.raw_shape = {k: tuple(int(d) for d in raw_shape[k]) for k in layout}
        self.capacity, self.device = int(capacity_rows), device
        self.patch_capacity = max(1, self.capacity // 8) if patch_rows is None else max(1, int(patch_rows))
        self.storage: Dict[str, torch.Tensor] = {}
        self.mean: Dict[str, torch.Tensor] = {}
        self.rstd: Dict[str, torch.Tensor] = {}
        self.dtype: Dict[str, torch.dtype] = {}
        self._ring = _Circular(self.capacity)  # rows staged by rollouts: findable by sequence number until lapped
        self._patch = _Circular(self.patch_capacity)  # rows staged by `bind` for one lease: storage slots capacity + ...
        self._lock = threading.Lock()
        self._leases = []
        # Events recorded on the readers' streams at release; the next WRITER of a region waits for them before it stages
        # rows there.  One list per region: the ring's writer is the inference stream, the patch area's writer is the
        # trainer's own stream (where waiting on its own event orders nothing) -- with one shared list whichever allocator
        # came first took the events away from the other, and a `put` could overwrite rows a released step still read.
        self._release_events = {"ring": [], "patch": []}
        self._write_events = {}  # writer stream -> event after its last put
        # A stamp = generation << GEN_SHIFT | (sequence number + 1): 0 -- what a zero-filled `analyzed_result` of a step that
        # never went through inference holds -- and negatives are dead by construction, and a stamp of another ring (a
        # recreated ring, another policy worker's) does not decode in this one.
        self.generation = int(next(_GENERATIONS))
        self._stamp_base = (self.generation << self.GEN_SHIFT) + 1
        self.stats = dict(rows_put=0, rows_bound=0, rows_patched=0, binds=0, binds_failed=0, puts_unstaged=0)

    @classmethod
    def for_policy(cls, policy, capacity_rows: int, patch_rows: Optional[int] = None) -> "ObsRing":
        net = policy.net
        return cls(net.obs_stage_layout(), net.obs_raw_shapes(), capacity_rows, policy.device, patch_rows)

    def keys(self):
        return self.layout.keys()

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.storage.values())

    @property
    def head(self) -> int:
        return self._ring.head

    # ------------------------------------------------------------------ storage
    def _staged_row_shape(self, key):
        lay, raw = self.layout[key], self.raw_shape[key]
        if lay[0] == "s2d":
            c, h, w = raw
            b = int(lay[1])
            return (h // b, w // b, c * b * b)
        return raw

    def _ensure(self, key, dtype):
        if key in self.storage:
            if self.dtype[key] != dtype:
                raise hip.HipError(f"observation `{key}`: ring holds {self.dtype[key]}, got {dtype}")
            return
        rows = self.capacity + self.patch_capacity
        self.dtype[key] = dtype
        self.storage[key] = torch.empty((rows, *self._staged_row_shape(key)), dtype=dtype, device=self.device)
        if self.layout[key][0] == "s2d":
            self.mean[key] = torch.empty(rows, dtype=torch.float32, device=self.device)
            self.rstd[key] = torch.empty(rows, dtype=torch.float32, device=self.device)

    def _check_rows(self, obs):
        n = None
        for k in self.layout:
            if k not in obs:
                raise KeyError(f"observation key `{k}` missing from the batch (has {list(obs)})")
            t = obs[k]
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
                raise hip.HipError(f"observation `{k}`: the ring stages contiguous device rows")
            if tuple(t.shape[1:]) != self.raw_shape[k]:
                raise hip.HipError(f"observation `{k}`: rows of shape {tuple(t.shape[1:])}, expected {self.raw_shape[k]}")
            n = t.shape[0] if n is None else n
            if t.shape[0] != n:
                raise hip.HipError("observation keys disagree on the number of rows")
            self._ensure(k, t.dtype)
        return n

    def _stage(self, obs, s0, n, keys=None):
        """Write ``n`` rows into storage slots [s0, s0 + n) in each key's layout, on the current stream."""
        for k, lay in self.layout.items():
            if keys is not None and k not in keys:
                continue
            t, store = obs[k], self.storage[k]
            if lay[0] == "s2d":
                c, h, w = self.raw_shape[k]
                row = store[0].numel() * store.element_size()
                hip.obs_space_to_depth(t.data_ptr(), t.dtype == torch.uint8, n, c, h, w, int(lay[1]), store.data_ptr() + s0 * row,
                                       self.mean[k].data_ptr() + 4 * s0, self.rstd[k].data_ptr() + 4 * s0)
            else:
                store[s0:s0 + n].copy_(t)

    def _decode(self, refs: np.ndarray):
        """Stamps -> (sequence numbers, mask of stamps that are this ring's at all)."""
        mine = (refs >> self.GEN_SHIFT) == self.generation
        low = refs & self.SEQ_MASK
        return low - 1, mine & (low >= 1)

    def _wait_released(self, events):
        if events:  # readers that released earlier may still be running on their stream
            stream = torch.cuda.current_stream(self.device)
            for ev in events:
                stream.wait_event(ev)

    # ------------------------------------------------------------------ producer side (rollout)
    def _alloc(self, n: int, check=None) -> int:
        """First sequence number of a run of ``n`` ring slots (never laps a leased row).  ``check(seq)`` runs under the lock
        before the run is taken; if it raises, the head has not moved."""
        with self._lock:
            floor = min((l.min_seq for l in self._leases if l.min_seq is not None), default=None)
            seq = self._ring.alloc(n, floor, "observation ring", check)
            events, self._release_events["ring"] = self._release_events["ring"], []
        self._wait_released(events)
        return seq

    def put(self, obs: Dict[str, torch.Tensor]):
        """Stage one inference batch.  ``obs``: key -> device rows ``[n, *raw_shape]`` (uint8 or float32) for every key of
        the ring.  Returns (stamps int64 numpy ``[n]``, key -> ``RingObs`` over the new run of slots).  Raises BufferError
        when the run would lap rows a training step has leased (``put_or_skip`` for callers that must not fail)."""
        n = self._check_rows(obs)
        seq = self._alloc(n)
        s0 = seq % self.capacity
        self._stage(obs, s0, n)
        stream = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(stream)
        with self._lock:
            self._write_events[stream.cuda_stream] = ev
            self.stats["rows_put"] += n
        return np.arange(seq, seq + n, dtype=np.int64) + self._stamp_base, {k: RingObs(self, k, (n,), span=s0) for k in self.layout}

    def stackable(self, key: str) -> bool:
        """Whether ``put_stacked`` can assemble this key's rows from single planes: uint8 frame stacks [C, H, W] staged in the
        block-4 space-to-depth layout."""
        lay, raw = self.layout[key], self.raw_shape[key]
        return lay[0] == "s2d" and int(lay[1]) == 4 and len(raw) == 3 and raw[1] % 4 == 0 and raw[2] % 4 == 0 and raw[1] * raw[2] % 16 == 0

    def put_stacked(self, planes: Dict[str, torch.Tensor], prev, full: Optional[Dict[str, torch.Tensor]] = None):
        """Stage one inference batch of FRAME-STACKED observations from their newest planes (the reference's `FrameStack`,
        atari_wrappers.py:211-242: the C latest frames, newest last; reset = C copies).  ``planes``: key -> device uint8
        ``[n, H, W]`` (or ``[n, 1, H, W]``), the newest frame of each row; ``prev``: int64 ``[n]`` -- the stamp this ring returned
        for the SAME environment's previous observation, or 0 at an episode start (the stack is then C copies of the plane);
        ``full``: the ring's other keys, whole rows as for ``put``.  Row i becomes [channels 1.. of row prev[i], planes[i]] and
        is, byte for byte and statistic for statistic, what ``put`` would have staged from the whole stack -- for a quarter of
        the bytes over the host link.  Raises LookupError when a previous observation is no longer in the ring (the caller then
        sends whole stacks through ``put``), BufferError as ``put``."""
        full = full or {}
        for k in self.layout:
            if (k in planes) == (k in full):
                raise KeyError(f"observation key `{k}`: exactly one of `planes` / `full` must hold it")
            if k in planes and not self.stackable(k):
                raise hip.HipError(f"observation `{k}` is not a uint8 frame stack in the block-4 space-to-depth layout")
        tensors = {}
        for k, t in planes.items():
            c, h, w = self.raw_shape[k]
            if t.dim() == 4 and t.shape[1] == 1:
                t = t[:, 0]
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous() and t.dtype == torch.uint8 and tuple(t.shape[1:]) == (h, w)):
                raise hip.HipError(f"observation `{k}`: the newest planes must be contiguous device uint8 [n, {h}, {w}]")
            tensors[k] = t
            self._ensure(k, torch.uint8)
        n = next(iter(tensors.values())).shape[0]
        if any(t.shape[0] != n for t in tensors.values()) or (full and self._check_rows_subset(full) != n):
            raise hip.HipError("observation keys disagree on the number of rows")
        prev = np.asarray(prev, dtype=np.int64).reshape(-1)
        if prev.shape[0] != n:
            raise hip.HipError("`prev`: one stamp per row")
        pseq, mine = self._decode(prev)
        fresh = prev == 0

        def predecessors_survive(seq):
            # validated against the run this call WOULD take, before it takes it: a refused call burns no slots and laps nobody
            # else's predecessors (a dead stamp -- a lapped row, another ring's stamp, the -1 of `put_or_skip` -- is refused here)
            ok = fresh | (mine & (pseq < seq) & (pseq + self.capacity >= seq + n))  # still there once this run is written
            if not ok.all():
                raise LookupError(f"{int((~ok).sum())} of {n} previous observations are no longer in the ring: send whole stacks")

        seq = self._alloc(n, predecessors_survive)
        s0 = seq % self.capacity
        slots = torch.from_numpy(np.where(fresh, -1, pseq % self.capacity).astype(np.int32)).to(self.device, non_blocking=True)
        for k, t in tensors.items():
            c, h, w = self.raw_shape[k]
            hip.ring_stack_push(self.storage[k], t, slots, s0, c, h, w, self.mean[k], self.rstd[k])
        if full:
            self._stage(full, s0, n, keys=list(full))
        stream = torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(stream)
        with self._lock:
            self._write_events[stream.cuda_stream] = ev
            self.stats["rows_put"] += n
            self.stats["rows_put_stacked"] = self.stats.get("rows_put_stacked", 0) + n
        return np.arange(seq, seq + n, dtype=np.int64) + self._stamp_base, {k: RingObs(self, k, (n,), span=s0) for k in self.layout}

    def _check_rows_subset(self, obs):
        n = None
        for k, t in obs.items():
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()) or tuple(t.shape[1:]) != self.raw_shape[k]:
                raise hip.HipError(f"observation `{k}`: the ring stages contiguous device rows of shape {self.raw_shape[k]}")
            n = t.shape[0] if n is None else n
            if t.shape[0] != n:
                raise hip.HipError("observation keys disagree on the number of rows")
            self._ensure(k, t.dtype)
        return n

    def put_or_skip(self, obs: Dict[str, torch.Tensor]):
        """``put``, or -- when the ring is full of leased rows (rollouts running beside a training step that holds a lease)
        -- nothing staged: (stamps of -1, None).  The forward pass then reads the caller's own rows, and the trainer uploads
        these rows from the sample's host copy when it binds it (a dead stamp)."""
        try:
            return self.put(obs)
        except BufferError:
            n = next(iter(obs.values())).shape[0]
            with self._lock:
                self.stats["puts_unstaged"] += 1
            return np.full(n, -1, dtype=np.int64), None

    # ------------------------------------------------------------------ consumer side (trainer)
    def alive(self, refs: np.ndarray) -> np.ndarray:
        with self._lock:
            head = self._ring.head
        seq, mine = self._decode(np.asarray(refs, dtype=np.int64))
        return mine & (seq < head) & (seq + self.capacity >= head)

    def _patch_rows(self, lease: ObsLease, rows: Dict[str, torch.Tensor]) -> np.ndarray:
        """Stage rows that are not alive in the ring into the patch area, for this lease only; returns their storage slots."""
        n = self._check_rows(rows)
        with self._lock:
            floor = min((l.patch_start for l in self._leases if l.patch_start is not None), default=None)
            if floor is None and lease.patch_start is not None:
                floor = lease.patch_start
            seq = self._patch.alloc(n, floor, "observation ring patch area")
            if lease.patch_start is None:
                lease.patch_start = seq
                if lease not in self._leases:
                    self._leases.append(lease)
            events, self._release_events["patch"] = self._release_events["patch"], []
        self._wait_released(events)
        s0 = self.capacity + seq % self.patch_capacity
        self._stage(rows, s0, n)
        return np.arange(s0, s0 + n, dtype=np.int64)

    def bind(self, refs, host_obs: Optional[Dict[str, "torch.Tensor"]] = None, piece_rows: int = 4096,
             refs_device: Optional[torch.Tensor] = None):
        """``refs``: int64 ``[Tb, B]`` (or ``[Tb, B, 1]``) sequence numbers from the sample, or None (no references: every
        row is uploaded).  ``host_obs``: key -> host rows ``[Tb, B, *raw_shape]`` (pinned torch tensors or numpy), used for
        the rows whose reference is not alive: they are uploaded into the patch area.  ``refs_device``: the same stamps
        already on the device (the sample's own leaf): with every stamp alive the slot index is then computed there and
        the host only takes their minimum and maximum.  Returns ``(key -> RingObs [Tb, B, ...], lease)``, or None when
        the ring cannot serve the sample (a dead reference without a host copy, or more dead rows than the patch area
        holds): the caller then copies the observations itself."""
        keys = list(self.layout)
        if refs is None:
            if not host_obs:
                return None
            lead = tuple(next(iter(host_obs.values())).shape[:2])
            refs = np.full(lead, -1, dtype=np.int64)
        refs = np.asarray(refs)
        if refs.ndim == 3 and refs.shape[2] == 1:
            refs = refs[..., 0]
        if refs.dtype != np.int64:
            refs = refs.astype(np.int64)
        # every stamp this ring's own and alive <=> min and max carry its generation and their sequence numbers are alive
        lo, hi = (int(refs.min()), int(refs.max())) if refs.size else (self._stamp_base, self._stamp_base - 1)
        same_gen = (lo >> self.GEN_SHIFT) == self.generation and (hi >> self.GEN_SHIFT) == self.generation and (lo & self.SEQ_MASK) >= 1
        lo, hi = (lo & self.SEQ_MASK) - 1, (hi & self.SEQ_MASK) - 1
        with self._lock:  # liveness and the lease in one step: no allocation can slip in between
            head = self._ring.head
            if same_gen and hi < head and lo + self.capacity >= head:  # the common case: every stamp alive
                lease = ObsLease(self, lo)
                self._leases.append(lease)
                ok = None
            else:
                seq, mine = self._decode(refs)
                ok = mine & (seq < head) & (seq + self.capacity >= head)
                lease = ObsLease(self, int(seq[ok].min()) if ok.any() else None)
                if lease.min_seq is not None:
                    self._leases.append(lease)
        stream = torch.cuda.current_stream(self.device)
        if ok is None:
            if refs_device is not None and refs_device.is_cuda and refs_device.dtype == torch.int64 and refs_device.is_contiguous():
                index = torch.empty(refs.shape, dtype=torch.int32, device=self.device)
                hip.ring_slots(refs_device, self.capacity, index, base=self._stamp_base)
            else:
                index = torch.from_numpy(((refs - self._stamp_base) % self.capacity).astype(np.int32)).to(self.device, non_blocking=True)
            return self._bound(keys, refs.shape, index, lease, stream, 0)
        slots = np.where(ok, seq % self.capacity, -1)
        patched = 0
        try:
            if not ok.all():
                if not host_obs or any(k not in host_obs for k in keys):
                    raise LookupError("dead observation references and no host copy")
                if int((~ok).sum()) > self.patch_capacity:
                    raise BufferError("more dead rows than the patch area holds")
                host = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))) for k, v in host_obs.items()}
                for t in range(refs.shape[0]):
                    miss = ~ok[t]
                    if not miss.any():
                        continue
                    cols = None if miss.all() else torch.from_numpy(np.nonzero(miss)[0])
                    m = refs.shape[1] if cols is None else int(cols.numel())
                    for c0 in range(0, m, piece_rows):
                        c1 = min(m, c0 + piece_rows)
                        rows = {}
                        for k in keys:
                            src = host[k][t]
                            src = src[c0:c1] if cols is None else src[cols[c0:c1]]
                            if src.dtype == torch.bool:
                                src = src.view(torch.uint8)
                            rows[k] = src.to(self.device, non_blocking=True).contiguous()
                        got = self._patch_rows(lease, rows)
                        if cols is None:
                            slots[t, c0:c1] = got
                        else:
                            slots[t, cols[c0:c1].numpy()] = got
                        patched += c1 - c0
        except (BufferError, LookupError):
            self.release(lease, record=False)
            with self._lock:
                self.stats["binds_failed"] += 1
            return None
        index = torch.from_numpy(slots.astype(np.int32)).to(self.device, non_blocking=True)
        if patched:
            lease.uploaded = torch.cuda.Event()
            lease.uploaded.record(stream)
        return self._bound(keys, refs.shape, index, lease, stream, patched)

    def _bound(self, keys, lead, index, lease, stream, patched):
        with self._lock:
            events = [ev for s, ev in self._write_events.items() if s != stream.cuda_stream]
            self.stats["rows_bound"] += int(np.prod(lead))
            self.stats["rows_patched"] += patched
            self.stats["binds"] += 1
        for ev in events:  # rows written on the inference thread's stream
            stream.wait_event(ev)
        return {k: RingObs(self, k, lead, index=index) for k in keys}, lease

    def release(self, lease: Optional[ObsLease], record: bool = True):
        """The step that read the leased rows has been enqueued: later allocations may lap them (after that work)."""
        if lease is None:
            return
        ev = None
        if record and torch.cuda.is_available():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
        with self._lock:
            if lease in self._leases:
                self._leases.remove(lease)
            if ev is not None:
                self._release_events["ring"].append(ev)
                self._release_events["patch"].append(ev)
