"""Small data carriers of the PPO path (mirrors of reference types).

``PPORolloutAnalyzedResult``: ``actor_critic_policy.py:22-25`` (namedarray stored in every sample).
``SampleAnalyzedResult``:     ``mappo.py:21-33`` (what ``policy.analyze(target="ppo")`` returns).
"""
import dataclasses
from typing import Optional

import torch

from srl_amd.api.policy import AnalyzedResult
from srl_amd.namedarray import NamedArray


class PPORolloutAnalyzedResult(AnalyzedResult, NamedArray):

    def __init__(self, log_probs, value, adv=None, ret=None, obs_ref=None):
        """``obs_ref``: int64 ``[.., 1]`` reference of the step's observation in the policy's HBM observation ring
        (runtime/obs_ring.py), present only when the rollout ran with a ring attached -- the field is absent otherwise,
        so the wire formats are the reference's."""
        extra = {} if obs_ref is None else dict(obs_ref=obs_ref)
        super().__init__(log_probs=log_probs, value=value, adv=adv, ret=ret, **extra)


@dataclasses.dataclass
class SampleAnalyzedResult:
    old_action_log_probs: torch.Tensor  # [T, B, 1]
    new_action_log_probs: torch.Tensor  # [T, B, 1]
    state_values: torch.Tensor  # [T, B, value_dim]
    entropy: Optional[torch.Tensor] = None  # [T, B, 1]
