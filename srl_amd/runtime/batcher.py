"""Inference batcher of the policy worker: many small ``RolloutRequest`` s -> one batched ``policy.rollout`` call.

Counterpart of ``PolicyWorker._batch_step`` / ``_inference`` (reference
``distributed/system/policy_worker.py:162-242``; local variant ``local/system/basic/policy_worker.py:144-217``):

* requests are concatenated on axis 0; a batch that was queued but whose inference has not started is
  merged with the newcomers instead of starting a second batch (``:213-219``);
* at most ``batch_size`` rows go out per call, the remainder is carried to the next batch (``:221-226``);
* fresh parameters are swapped in before the batch runs (``:166-172``): here ``parameter_source()`` returns a
  checkpoint or ``None``;
* the response carries ``client_id / request_id / received_time / buffer_index`` of its requests plus
  ``policy_name`` and ``policy_version_steps = policy.version`` (``:181-188``).

Transport (ZMQ / shared memory streams) and threads are out of scope: this is the
synchronous core the reference's threads drive.
"""
from typing import Callable, List, Optional

import numpy as np

from srl_amd.api import policy as policy_api
from srl_amd.namedarray import recursive_aggregate


class InferenceBatcher:
    """``stage_on_device``: where the policy lives on a GPU, the observation leaves of a batch are never
    concatenated on the host (the reference's ``np.concatenate`` of the frames is the largest cost of its batcher,
    SURVEY.md 8a-11): each request's leaves are copied H2D straight into their rows of one reusable device block,
    in their wire dtype, and ``policy.rollout`` reads that block.  The small id / flag leaves are still folded on
    the host.  Default: on when the policy's device is a GPU."""

    def __init__(self, policy: policy_api.Policy, policy_name: str = "default", batch_size: int = 10240,
                 parameter_source: Optional[Callable[[], Optional[dict]]] = None, stage_on_device: Optional[bool] = None):
        self.policy = policy
        self.policy_name = policy_name
        self.batch_size = batch_size
        self.parameter_source = parameter_source
        self._incoming: List[policy_api.RolloutRequest] = []
        self._queued: List[policy_api.RolloutRequest] = []  # the formed, not yet run batch, as its parts
        dev = str(getattr(policy, "device", "cpu"))
        self._device = dev
        self.stage_on_device = (dev != "cpu") if stage_on_device is None else bool(stage_on_device)
        self._blocks = {}  # obs key -> reusable device block [>= batch rows, ...]

    def post(self, request: policy_api.RolloutRequest):
        self._incoming.append(request)

    def pending_rows(self) -> int:
        return sum(r.length(dim=0) for r in self._incoming) + sum(r.length(dim=0) for r in self._queued)

    def batch_step(self) -> int:
        """Fold incoming requests (and a not-yet-started batch) into the next batch; returns #requests folded."""
        if not self._incoming:
            return 0
        taken, self._incoming = self._incoming, []
        parts, rows, keep = self._queued + taken, 0, []
        for i, part in enumerate(parts):
            n = part.length(dim=0)
            if rows + n <= self.batch_size:
                keep.append(part)
                rows += n
                continue
            room = self.batch_size - rows  # the batch is capped; the remainder is carried (slices are views)
            if room > 0:
                keep.append(part[:room])
            self._incoming = [part[room:]] + parts[i + 1:]
            break
        self._queued = keep
        return len(taken)

    def _fold(self, parts: List[policy_api.RolloutRequest]) -> policy_api.RolloutRequest:
        if not self.stage_on_device:
            return recursive_aggregate(parts, lambda xs: np.concatenate(xs, axis=0))
        import torch
        rows = sum(p.length(dim=0) for p in parts)
        obs = {}
        for key, first in parts[0].obs.items():
            if first is None:
                obs[key] = None
                continue
            first = np.asarray(first)
            blk = self._blocks.get(key)
            dtype = torch.from_numpy(first[:0].view(np.uint8) if first.dtype == np.bool_ else first[:0]).dtype
            if blk is None or blk.shape[0] < rows or blk.shape[1:] != first.shape[1:] or blk.dtype != dtype:
                blk = torch.empty((max(rows, self.batch_size), *first.shape[1:]), dtype=dtype, device=self._device)
                self._blocks[key] = blk
            r0 = 0
            for p in parts:
                leaf = np.ascontiguousarray(np.asarray(p.obs[key]))
                if leaf.dtype == np.bool_:
                    leaf = leaf.view(np.uint8)
                blk[r0:r0 + leaf.shape[0]].copy_(torch.from_numpy(leaf), non_blocking=True)
                r0 += leaf.shape[0]
            obs[key] = blk[:rows]
        stripped = []
        for p in parts:  # everything but the observations is small: fold on the host as the reference does
            q = policy_api.RolloutRequest(**{k: (None if k == "obs" else v) for k, v in p.items()})
            stripped.append(q)
        out = recursive_aggregate(stripped, lambda xs: np.concatenate(xs, axis=0)) if len(stripped) > 1 else stripped[0]
        out.obs = type(parts[0].obs)(**obs)
        return out

    def inference(self) -> Optional[policy_api.RolloutResult]:
        """Run the queued batch (if any) and stamp the response."""
        if not self._queued:
            return None
        parts, self._queued = self._queued, []
        requests = self._fold(parts)
        if self.parameter_source is not None:
            ckpt = self.parameter_source()
            if ckpt is not None:
                self.policy.load_checkpoint(ckpt)
        responses = self.policy.rollout(requests)
        shape = requests.client_id.shape
        responses.client_id = requests.client_id
        responses.request_id = requests.request_id
        responses.received_time = requests.received_time
        responses.buffer_index = requests.buffer_index
        responses.ready = np.full(shape=shape, fill_value=True)
        responses.policy_name = np.full(shape=shape, fill_value=self.policy_name)
        responses.policy_version_steps = np.full(shape=shape, fill_value=self.policy.version)
        return responses

    def poll(self) -> List[policy_api.RolloutResult]:
        """Drain everything that is pending, ``batch_size`` rows at a time."""
        out = []
        while self._incoming or self._queued:
            self.batch_step()
            res = self.inference()
            if res is not None:
                out.append(res)
        return out
