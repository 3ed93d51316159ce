"""Trainer plugin interface (mirror of reference ``api/trainer.py:14-264``).

``SampleBatch`` has the reference's exact field set (``:14-82``): the trainer worker stacks
``batch_size`` per-agent ``[Tb, ...]`` samples on axis 1 into ``[Tb, B, ...]`` leaves
(``base/buffer.py:120-121``) and hands that to ``Trainer.step``.
"""
import dataclasses
from abc import ABC
from typing import Dict, List, Optional, Union

import numpy as np
import torch
import torch.distributed as dist

from srl_amd.api import config
from srl_amd.api import policy as policy_api
from srl_amd.api.environment import Action
from srl_amd.namedarray import NamedArray


class SampleBatch(NamedArray):
    """General sample container used by every algorithm (unused entries stay ``None``)."""

    def __init__(self,
                 obs: NamedArray,
                 on_reset: np.ndarray = None,
                 done: np.ndarray = None,
                 truncated: np.ndarray = None,
                 action: Action = None,
                 reward: np.ndarray = None,
                 info: NamedArray = None,
                 info_mask: np.ndarray = None,
                 policy_state: policy_api.PolicyState = None,
                 analyzed_result: policy_api.AnalyzedResult = None,
                 policy_name: np.ndarray = None,
                 policy_version_steps: np.ndarray = None,
                 actor_worker_post_timestamp: np.ndarray = None,
                 actor_worker_flush_timestamp: np.ndarray = None,
                 trainer_worker_recv_timestamp: np.ndarray = None,
                 trainer_worker_batch_timestamp: np.ndarray = None,
                 send_timestamp: np.ndarray = None,
                 buffer_recv_timestamp: np.ndarray = None,
                 sampling_weight: np.ndarray = None,
                 **kwargs):
        super().__init__(
            obs=obs,
            on_reset=on_reset,
            done=done,
            truncated=truncated,
            action=action,
            reward=reward,
            info=info,
            info_mask=info_mask,
            policy_state=policy_state,
            analyzed_result=analyzed_result,
            policy_name=policy_name,
            policy_version_steps=policy_version_steps,
            send_timestamp=send_timestamp,
            buffer_recv_timestamp=buffer_recv_timestamp,
            actor_worker_post_timestamp=actor_worker_post_timestamp,
            actor_worker_flush_timestamp=actor_worker_flush_timestamp,
            trainer_worker_recv_timestamp=trainer_worker_recv_timestamp,
            trainer_worker_batch_timestamp=trainer_worker_batch_timestamp,
        )
        self.register_metadata(sampling_weight=sampling_weight)


class TrajPostprocessor(ABC):
    """Post-processes a finished trajectory on the actor before it is sent (e.g. GAE)."""

    def process(self, memory: List[SampleBatch]):
        raise NotImplementedError()


class NullTrajPostprocessor(TrajPostprocessor):

    def process(self, memory: List[SampleBatch]):
        return memory


@dataclasses.dataclass
class TrainerStepResult:
    stats: Dict  # plain python numbers, logged / pickled by the runtime
    step: int  # policy version after the step
    agree_pushing: Optional[bool] = True
    priorities: Optional[np.ndarray] = None


class Trainer:

    @property
    def policy(self) -> policy_api.Policy:
        raise NotImplementedError()

    def step(self, samples: SampleBatch) -> TrainerStepResult:
        raise NotImplementedError()

    def distributed(self, **kwargs):
        raise NotImplementedError()

    def get_checkpoint(self, *args, **kwargs):
        raise NotImplementedError()

    def load_checkpoint(self, checkpoint, **kwargs):
        raise NotImplementedError()


class PytorchTrainer(Trainer, ABC):
    """Trainer whose tensors live in PyTorch(-ROCm) memory; one process per GPU."""

    @property
    def policy(self) -> policy_api.Policy:
        return self._policy

    def __init__(self, policy: policy_api.Policy):
        if policy.device != "cpu":
            torch.cuda.set_device(policy.device)
        self._policy = policy

    def distributed(self, rank, world_size, init_method, **kwargs):
        """Join the data-parallel group: backend "nccl" (= RCCL over xGMI on ROCm) on GPU, gloo on CPU.

        Same call signature as the reference (``api/trainer.py:179-189``); an already initialised
        default group (torchrun) is reused.
        """
        if not dist.is_initialized():
            on_gpu = self.policy.device != "cpu" and torch.cuda.is_available() and dist.is_nccl_available()
            dist.init_process_group(backend="nccl" if on_gpu else "gloo",
                                    init_method=init_method,
                                    rank=rank,
                                    world_size=world_size)
        self.policy.distributed()


ALL_TRAINER_CLASSES = {}


def register(name, trainer_class):
    ALL_TRAINER_CLASSES[name] = trainer_class


def make(cfg: Union[str, config.Trainer], policy_cfg: Union[str, config.Policy]) -> Trainer:
    if isinstance(cfg, str):
        cfg = config.Trainer(type_=cfg)
    if isinstance(policy_cfg, str):
        policy_cfg = config.Policy(type_=policy_cfg)
    cls = ALL_TRAINER_CLASSES[cfg.type_]
    policy = policy_api.make(policy_cfg)
    policy.train_mode()
    return cls(policy=policy, **cfg.args)


ALL_TRAJ_POSTPROCESSOR_CLASSES = {}


def register_traj_postprocessor(name, cls_):
    ALL_TRAJ_POSTPROCESSOR_CLASSES[name] = cls_


register_traj_postprocessor('null', NullTrajPostprocessor)


def make_traj_postprocessor(cfg: Union[str, config.TrajPostprocessor]):
    if isinstance(cfg, str):
        cfg = config.TrajPostprocessor(cfg)
    return ALL_TRAJ_POSTPROCESSOR_CLASSES[cfg.type_](**cfg.args)
