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

    def __init__(self, policy: policy_api.Policy, policy_name: str = "default", batch_size: int = 10240,
                 parameter_source: Optional[Callable[[], Optional[dict]]] = None):
        self.policy = policy
        self.policy_name = policy_name
        self.batch_size = batch_size
        self.parameter_source = parameter_source
        self._incoming: List[policy_api.RolloutRequest] = []
        self._queued: Optional[policy_api.RolloutRequest] = None  # formed but not yet run

    def post(self, request: policy_api.RolloutRequest):
        self._incoming.append(request)

    def pending_rows(self) -> int:
        rows = sum(r.length(dim=0) for r in self._incoming)
        return rows + (self._queued.length(dim=0) if self._queued is not None else 0)

    def batch_step(self) -> int:
        """Fold incoming requests (and a not-yet-started batch) into the next batch; returns #requests folded."""
        if not self._incoming:
            return 0
        taken, self._incoming = self._incoming, []
        parts = ([self._queued] if self._queued is not None else []) + taken
        agg = recursive_aggregate(parts, lambda xs: np.concatenate(xs, axis=0))
        if agg.length(dim=0) > self.batch_size:
            self._queued = agg[:self.batch_size]
            self._incoming = [agg[self.batch_size:]]
        else:
            self._queued = agg
        return len(taken)

    def inference(self) -> Optional[policy_api.RolloutResult]:
        """Run the queued batch (if any) and stamp the response."""
        if self._queued is None:
            return None
        requests, self._queued = self._queued, None
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
        while self._incoming or self._queued is not None:
            self.batch_step()
            res = self.inference()
            if res is not None:
                out.append(res)
        return out
