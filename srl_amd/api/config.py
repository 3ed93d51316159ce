"""Configuration dataclasses for the hot path (subset of reference ``api/config.py``).

Only the types that name plugins on the rollout -> GAE -> PPO path are mirrored:
``Environment`` / ``Policy`` / ``Trainer`` / ``TrajPostprocessor`` / ``DataAugmenter`` (``type_`` + free-form
``args``; reference ``api/config.py:35-62``), the per-agent sample contract ``AgentSpec``
(``:329-354``) and the two worker configs whose fields the batcher and the trainer loop read
(``PolicyWorker`` ``:404-420``, ``TrainerWorker`` ``:423-449``). Streams, parameter DBs, schedulers and
the experiment registry are control plane and out of scope.
"""
import dataclasses
import math
from typing import Any, Dict, List, Optional, Union


@dataclasses.dataclass
class Environment:
    type_: str
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class DataAugmenter:
    type_: str
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class Policy:
    type_: str
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)
    init_ckpt_dir: Optional[str] = None


@dataclasses.dataclass
class Trainer:
    type_: str
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class TrajPostprocessor:
    type_: str
    args: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class AgentSpec:
    """Per-agent sample contract (reference ``api/config.py:329-354``)."""
    index_regex: str = ".*"
    inference_stream_idx: int = 0
    sample_stream_idx: Union[int, List[int]] = 0
    sample_steps: int = 200
    bootstrap_steps: int = 1
    burn_in_steps: int = 0
    deterministic_action: bool = False
    send_after_done: bool = False
    send_full_trajectory: bool = False
    pad_trajectory: bool = False
    trajectory_postprocessor: Union[str, TrajPostprocessor] = 'null'
    compute_gae_before_send: bool = False  # dead config in the reference; kept for drop-in
    gae_args: Optional[Dict] = dataclasses.field(default_factory=dict)
    send_concise_info: bool = False
    update_concise_step: bool = False
    stack_frames: int = 0

    def __post_init__(self):
        if (not self.send_after_done and not self.send_full_trajectory
                and self.trajectory_postprocessor != "null"):
            raise ValueError("Either `send_after_done` or `send_full_trajectory` should be True if "
                             "trajectory postprocessor is activated!")


@dataclasses.dataclass
class PolicyWorker:
    """Fields of the reference's policy-worker config that the inference batcher reads."""
    policy_name: str
    policy: Union[str, Policy]
    batch_size: int = 10240  # upper bound of one inference batch (reference :412)
    max_inference_delay: float = 0.1
    pull_frequency_seconds: float = 1


@dataclasses.dataclass
class TrainerWorker:
    """Fields of the reference's trainer-worker config that the trainer loop reads."""
    policy_name: str
    trainer: Union[str, Trainer]
    policy: Union[str, Policy]
    buffer_name: str = "priority_queue"
    buffer_args: Dict[str, Any] = dataclasses.field(default_factory=dict)
    push_frequency_seconds: Optional[float] = 1.
    push_frequency_steps: Optional[int] = 1
    preemption_steps: float = math.inf
    train_for_seconds: float = 365 * 24 * 3600
