"""Native RCCL communicator behind the C ABI (``srl_comm_*``, ``srl_allreduce_*``, ``srl_broadcast_params``).

One process per GPU.  The communicator is bootstrapped through the ``torch.distributed`` process group the trainer
already has (``trainer.distributed`` mirrors reference ``api/trainer.py:113-128``): rank 0 draws the RCCL unique id,
the 128 bytes travel by one broadcast, every rank joins with ``srl_comm_init``.  From then on the hot path's
collectives go through the C ABI on a dedicated side HIP stream, ordered against the compute stream with events:

* ``all_reduce_f64_async`` / ``all_reduce_f32_async`` enqueue on the side stream after everything the compute
  stream has enqueued so far (the data they reduce is ready), and return at once;
* ``join`` makes the compute stream wait for every collective enqueued so far.

So the 24-byte statistics all-reduce runs under the first chunk's forward pass and the gradient buckets run under
the rest of the backward pass, with no torch dispatcher work and no host synchronisation in between.
"""
import ctypes
import logging
import os
from typing import Optional

import torch
import torch.distributed as dist

from srl_amd import hip

logger = logging.getLogger("srl_amd.comm")


class NativeComm:

    def __init__(self, handle: int, rank: int, world: int, device: str):
        self._h = ctypes.c_void_p(handle)
        self.rank, self.world, self.device = rank, world, device
        self.stream = torch.cuda.Stream(device=device)
        self._pending = False

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_process_group(cls, device: str, group=None) -> Optional["NativeComm"]:
        """Communicator over the ranks of the (default) process group, or None when it is not asked for or cannot be formed
        (the caller then keeps using torch.distributed, whose nccl backend IS RCCL): ``SRL_COMM=native`` opts in.  The
        default stays ``torch`` until a multi-GPU run has validated the native path -- it has only ever seen one rank."""
        if os.environ.get("SRL_COMM", "torch") != "native":
            return None
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # a rank on which librccl does not resolve must say so BEFORE anyone enters ncclCommInitRank (which blocks until
        # every rank has joined): agree on availability over the existing process group first
        try:
            avail = int(hip.lib().srl_comm_available())
        except (hip.HipError, OSError, AttributeError):  # pragma: no cover - depends on the box
            avail = 0
        flag = torch.tensor([avail], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            logger.warning("librccl does not resolve on every rank: using torch.distributed collectives")
            return None
        comm = cls._try_init(device, group, rank, world)
        # every rank must take the same path: one that failed to join would otherwise wait in torch.distributed for peers
        # that sit in an RCCL collective
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0 and comm is not None:
            logger.warning("native RCCL communicator formed here but not on every rank: using torch.distributed collectives")
            comm.close()
            comm = None
        return comm

    @classmethod
    def _try_init(cls, device, group, rank, world) -> Optional["NativeComm"]:
        # the id broadcast happens on every rank whatever rank 0 managed to do (zeros = "no id"): a rank that raised before
        # it would leave its peers waiting in the broadcast
        idbuf = (ctypes.c_uint8 * 128)()
        if rank == 0:
            try:
                hip._check(hip.lib().srl_comm_unique_id(idbuf), "srl_comm_unique_id")
            except (hip.HipError, OSError, AttributeError) as e:  # pragma: no cover - depends on the box
                logger.warning("native RCCL communicator unavailable (%s): using torch.distributed collectives", e)
                idbuf = (ctypes.c_uint8 * 128)()
        t = torch.tensor(list(bytes(idbuf)), dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
        if not any(raw):
            return None
        try:
            handle = ctypes.c_void_p()
            with torch.cuda.device(device):
                hip._check(hip.lib().srl_comm_init(ctypes.byref(handle), raw, rank, world), "srl_comm_init")
            comm = cls(handle.value, rank, world, device)
            seen = ctypes.c_int(0)
            hip._check(hip.lib().srl_comm_world(comm._h, ctypes.byref(seen)), "srl_comm_world")
            if seen.value != world:
                raise hip.HipError(f"communicator has {seen.value} ranks, process group {world}")
            return comm
        except (hip.HipError, OSError, AttributeError) as e:  # pragma: no cover - depends on the box
            logger.warning("native RCCL communicator unavailable (%s): using torch.distributed collectives", e)
            return None

    def close(self):
        if self._h:
            torch.cuda.synchronize(self.device)
            hip._check(hip.lib().srl_comm_destroy(self._h), "srl_comm_destroy")
            self._h = None

    # ------------------------------------------------------------------ collectives
    def _enter(self):
        """The side stream picks up after the compute stream's current tail (the operands are produced there)."""
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        self._pending = True
        return self.stream.cuda_stream

    def all_reduce_f64_async(self, t: torch.Tensor):
        assert t.dtype == torch.float64 and t.is_contiguous() and t.is_cuda
        hip._check(hip.lib().srl_allreduce_stats_f64x3(self._enter(), self._h, t.data_ptr(), t.numel()),
                   "srl_allreduce_stats_f64x3")

    def all_reduce_f32_async(self, t: torch.Tensor):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        hip._check(hip.lib().srl_allreduce_grads(self._enter(), self._h, t.data_ptr(), t.numel()), "srl_allreduce_grads")

    def all_reduce_f32_inline(self, t: torch.Tensor):
        """The same all-reduce enqueued on the CURRENT stream, in its order (RCCL orders the collectives of one communicator by
        their enqueue order whatever streams they are on): for a reduction whose result the very next kernel needs."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        hip._check(hip.lib().srl_allreduce_grads(torch.cuda.current_stream(self.device).cuda_stream, self._h, t.data_ptr(), t.numel()),
                   "srl_allreduce_grads")

    def broadcast_async(self, t: torch.Tensor, root: int = 0):
        assert t.is_contiguous() and t.is_cuda
        hip._check(hip.lib().srl_broadcast_params(self._enter(), self._h, t.data_ptr(), t.numel() * t.element_size(), root),
                   "srl_broadcast_params")

    def join(self):
        """The compute stream waits for everything enqueued on the side stream so far (no host wait)."""
        if self._pending:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            self._pending = False
