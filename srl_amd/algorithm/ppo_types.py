"""Small data carriers of the PPO path (mirrors of reference types).

``PPORolloutAnalyzedResult``: ``actor_critic_policy.py:22-25`` (namedarray stored in every sample).
``SampleAnalyzedResult``:     ``mappo.py:21-33`` (what ``policy.analyze(target="ppo")`` returns).
``PPGPhase1AnalyzedResult`` / ``PPGPhase2AnalyzedResult`` / ``AuxiliaryStepResult``: ``phasic_policy_gradient.py:15-57``.
"""
import dataclasses
from typing import List, Optional

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


@dataclasses.dataclass
class PPGPhase1AnalyzedResult:
    """``policy.analyze(target="ppg_ppo_phase")`` (phasic_policy_gradient.py:15-29): the PPO quantities plus the auxiliary value."""
    old_action_log_probs: torch.Tensor  # [T, B, 1]
    new_action_log_probs: torch.Tensor  # [T, B, 1]
    aux_values: torch.Tensor  # [T, B, value_dim]; takes no gradient in the PPO phase
    state_values: torch.Tensor  # [T, B, value_dim]
    reward: torch.Tensor  # [T, B, value_dim or 1]
    entropy: Optional[torch.Tensor] = None  # [T, B, 1]


class CategoricalLogits:
    """What the auxiliary phase needs of a ``torch.distributions.Categorical`` (actor_critic_policy.py:266-270): ``logits`` are the
    NORMALISED log-probabilities [..., A] of one action head (as ``Categorical.logits``), ``probs`` their exponentials."""

    def __init__(self, logits: torch.Tensor):
        self.logits = logits

    @property
    def probs(self):
        return self.logits.exp()

    def detach_cpu(self):
        return CategoricalLogits(self.logits.detach().cpu())


@dataclasses.dataclass
class PPGPhase2AnalyzedResult:
    """``policy.analyze(target="ppg_aux_phase")`` (phasic_policy_gradient.py:32-44)."""
    action_dists: List[CategoricalLogits]  # one per action head, [T, B, A_h]
    auxiliary_value: torch.Tensor  # [T, B, value_dim], from the actor's features
    predicted_value: torch.Tensor  # [T, B, value_dim], from the critic head


@dataclasses.dataclass
class AuxiliaryStepResult:
    """phasic_policy_gradient.py:47-54."""
    auxiliary_value_loss: float
    value_head_loss: float
    policy_distance: float
