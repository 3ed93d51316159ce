"""Policy plugin interface (mirror of reference ``api/policy.py:14-302``).

``RolloutRequest`` / ``RolloutResult`` keep the reference's field sets and defaults (``:26-79``);
``Policy`` keeps the abstract method set (``:82-202``). Device selection differs by design: the
reference resolves one device from ``CUDA_VISIBLE_DEVICES`` through its cluster name service
(``:84-102``); here it is one process per GPU, device = ``cuda:$LOCAL_RANK`` (ROCm devices appear
under the ``cuda`` device type in PyTorch-ROCm) or ``"cpu"`` when no GPU is visible. On ``"cpu"`` a
policy can be built, check-pointed and inspected, but every compute entry point raises: there is no
CPU fallback on the product path.
"""
import logging
import os
from typing import Union

import numpy as np
import torch

from srl_amd.api import config
from srl_amd.api import environment
from srl_amd.namedarray import NamedArray

logger = logging.getLogger("Policy")


class PolicyState:
    pass


class AnalyzedResult:
    pass


class RolloutResult(NamedArray):

    def __init__(self,
                 action: environment.Action,
                 policy_state: PolicyState = None,
                 analyzed_result: AnalyzedResult = None,
                 client_id: np.ndarray = None,
                 request_id: np.ndarray = None,
                 received_time: np.ndarray = None,
                 policy_name: np.ndarray = None,
                 policy_version_steps: np.ndarray = None,
                 buffer_index: np.ndarray = None,
                 ready: np.ndarray = None,
                 **kwargs):
        super().__init__(action=action, policy_state=policy_state, analyzed_result=analyzed_result,
                         client_id=client_id, request_id=request_id, received_time=received_time,
                         policy_name=policy_name, policy_version_steps=policy_version_steps,
                         buffer_index=buffer_index, ready=ready, **kwargs)


class RolloutRequest(NamedArray):

    def __init__(self,
                 obs: NamedArray,
                 policy_state: PolicyState = None,
                 is_evaluation: np.ndarray = None,
                 on_reset: np.ndarray = None,
                 step_count: np.ndarray = None,
                 client_id: np.ndarray = None,
                 request_id: np.ndarray = None,
                 received_time: np.ndarray = None,
                 buffer_index: np.ndarray = None,
                 ready: np.ndarray = None,
                 **kwargs):
        # defaults as in the reference (:52-61); created per instance instead of shared module-level arrays
        def _d(v, default, dtype):
            return np.array([default], dtype=dtype) if v is None else v

        super().__init__(obs=obs,
                         policy_state=policy_state,
                         is_evaluation=_d(is_evaluation, False, np.uint8),
                         on_reset=_d(on_reset, False, np.uint8),
                         step_count=_d(step_count, -1, np.int32),
                         client_id=_d(client_id, -1, np.int32),
                         request_id=_d(request_id, -1, np.int32),
                         received_time=_d(received_time, -1, np.int64),
                         buffer_index=_d(buffer_index, -1, np.int32),
                         ready=_d(ready, False, np.bool_),
                         **kwargs)


def resolve_device() -> str:
    """One process per GPU: ``cuda:$LOCAL_RANK`` if a GPU is visible, else ``"cpu"``."""
    if torch.cuda.is_available():
        n = torch.cuda.device_count()
        return f"cuda:{int(os.environ.get('LOCAL_RANK', 0)) % max(n, 1)}"
    return "cpu"


class Policy:

    def __init__(self):
        self.device = resolve_device()
        logger.debug(f"Policy device pid {os.getpid()}: {self.device}")

    @property
    def default_policy_state(self):
        raise NotImplementedError()

    @property
    def version(self) -> int:
        raise NotImplementedError()

    @property
    def net(self):
        raise NotImplementedError()

    def analyze(self, sample, target, **kwargs):
        raise NotImplementedError()

    def reanalyze(self, sample, target, **kwargs):
        raise NotImplementedError()

    def rollout(self, requests: RolloutRequest, **kwargs) -> RolloutResult:
        raise NotImplementedError()

    def trace_by_sample_batch(self, sample):
        raise NotImplementedError()

    def parameters(self):
        raise NotImplementedError()

    def get_checkpoint(self):
        raise NotImplementedError()

    def load_checkpoint(self, checkpoint):
        raise NotImplementedError()

    def train_mode(self):
        raise NotImplementedError()

    def eval_mode(self):
        raise NotImplementedError()

    def inc_version(self):
        raise NotImplementedError()

    def distributed(self):
        raise NotImplementedError()


ALL_POLICY_CLASSES = {}


def register(name, policy_class):
    ALL_POLICY_CLASSES[name] = policy_class


def make(cfg: Union[str, config.Policy]) -> Policy:
    if isinstance(cfg, str):
        cfg = config.Policy(type_=cfg)
    return ALL_POLICY_CLASSES[cfg.type_](**cfg.args)
