"""``smac_rnn``: the multi-agent recurrent actor-critic of the StarCraft presets, on the HIP kernels.

Mirror of ``SMACPolicy`` (reference ``legacy/algorithm/ppo/game_policies/smac_rnn.py:170-410``): actor on
``obs.local_obs``, centralised critic on ``obs.state``, action masking with ``obs.available_action``, separate recurrent
states ``policy_state.{actor_hx, critic_hx}``, PopArt value head.  With ``shared=True`` one environment returns all
agents at once and every leaf carries an agent axis, ``[T, B, agents, ...]`` in samples and ``[N, agents, ...]`` in
rollout requests; the agents are folded into the batch axis for the network and unfolded on the way out (``:253-266``,
``:349-353``, ``:390-392``), and steps of dead agents (``obs.is_alive == 0``) get a new log-probability of ``-inf`` so
that no policy gradient flows through them (``:309-311``).

Two things differ from the reference file, both because it cannot run as shipped:

* **state width.**  ``SMACNet`` builds ``AutoResetRNN(hidden, hidden, num_layers)`` without an ``rnn_type``, so it gets
  that module's default, an LSTM (``autoreset_rnn.py:9``), whose state is ``cat(h, c)`` -- ``2 * hidden_dim`` wide.  The
  reference's ``default_policy_state`` is still ``hidden_dim`` wide (``:201-204``) and its ``rollout`` raises inside
  ``nn.LSTM`` ("Expected hidden[0] size ...").  Here the default state has the width the network needs.  The training
  side (``analyze`` and the trainer step on stored ``2H``-wide states) runs in the reference and is what the golden
  fixtures pin; the rollout fixture is generated with the reference's default state patched to ``2H`` in the harness.
* **shapes.**  The reference asks a live ``StarCraft2Env`` for the observation sizes (``smac_env.py:17-20``).  There is
  no StarCraft here: sizes come from the ``obs_shape / state_shape / act_dim / n_agents`` keywords, or from
  ``SMAC_SHAPES`` for the maps listed there (the standard SMAC feature sizes).

Agent-specific (attention) encoders are not on the HIP path.
"""
from typing import Optional

import numpy as np
import torch

from srl_amd import hip
from srl_amd.algorithm import netspec as ns
from srl_amd.algorithm.actor_critic import ActorCriticPolicy, to_device_leaf
from srl_amd.algorithm.ppo_types import PPORolloutAnalyzedResult
from srl_amd.api import policy as policy_api
from srl_amd.api.env_utils import DiscreteAction
from srl_amd.namedarray import NamedArray

# map -> (local_obs width, state width, #actions, #agents) with use_state_agent=True feature sets
SMAC_SHAPES = {"3m": (30, 48, 9, 3)}


class SMACAction(DiscreteAction):
    pass


class SMACPolicyState(NamedArray):

    def __init__(self, actor_hx: np.ndarray, critic_hx: np.ndarray):
        super().__init__(actor_hx=actor_hx, critic_hx=critic_hx)


def _width(shape):
    return int(shape[0]) if isinstance(shape, (tuple, list)) else int(shape)


class SMACPolicy(ActorCriticPolicy):

    @property
    def masks_dead_agents(self):  # only the shared path looks at obs.is_alive (smac_rnn.py:307-311)
        return self._shared

    def __init__(self,
                 map_name: Optional[str] = None,
                 hidden_dim: int = 64,
                 chunk_len: int = 10,
                 seed: int = 0,
                 shared: bool = False,
                 agent_specific_obs: bool = False,
                 agent_specific_state: bool = False,
                 act_init_gain: float = 0.01,
                 num_rnn_layers: int = 1,
                 denormalize_value_during_rollout: bool = False,
                 popart: bool = True,
                 unbiased_popart: bool = False,
                 popart_beta: float = 1 - 1e-5,
                 obs_shape=None,
                 state_shape=None,
                 act_dim: Optional[int] = None,
                 n_agents: Optional[int] = None,
                 **kwargs):
        policy_api.Policy.__init__(self)
        if agent_specific_obs or agent_specific_state:
            raise NotImplementedError("agent-specific (attention) SMAC encoders are not on the HIP path")
        if obs_shape is None or state_shape is None or act_dim is None or n_agents is None:
            if map_name not in SMAC_SHAPES:
                raise ValueError(f"SMAC map `{map_name}`: no StarCraft here to ask for the observation sizes; pass "
                                 f"obs_shape, state_shape, act_dim and n_agents (known maps: {sorted(SMAC_SHAPES)})")
            obs_shape, state_shape, act_dim, n_agents = SMAC_SHAPES[map_name]
        self.spec, init = ns.build_smac_netspec(_width(obs_shape), _width(state_shape), int(act_dim), hidden_dim,
                                                num_rnn_layers=num_rnn_layers, act_init_gain=act_init_gain, seed=seed)
        self._setup(init, chunk_len, seed, denormalize_value_during_rollout)
        self._popart_beta = float(popart_beta)
        self._popart_burn_in = 1000 if unbiased_popart else float("inf")  # smac_rnn.py:133-135
        self._use_popart = popart
        self._shared = shared
        self._n_agents = int(n_agents)

    @property
    def default_policy_state(self):
        L, W = self.spec.num_rnn_layers, self.spec.rnn_state_width
        if not L:
            return None
        shape = (self._n_agents, L, W) if self._shared else (L, W)
        return SMACPolicyState(np.zeros(shape, dtype=np.float32), np.zeros(shape, dtype=np.float32))

    def rollout(self, requests: policy_api.RolloutRequest, **kwargs) -> policy_api.RolloutResult:
        hip.require_gpu()
        fold = (lambda t: t.reshape(t.shape[0] * t.shape[1], *t.shape[2:])) if self._shared else (lambda t: t)
        obs = {k: fold(to_device_leaf(v, self.device, "obs")) for k, v in requests.obs.items() if v is not None}
        obs.pop("is_alive", None)
        bs = int(np.asarray(requests.on_reset).shape[0])
        n = bs * (self._n_agents if self._shared else 1)
        state = None
        L, W = self.spec.num_rnn_layers, self.spec.rnn_state_width
        if L:
            # the default (zero) state where the episode restarts, the carried one elsewhere (:354-361)
            keep = 1.0 - fold(to_device_leaf(requests.on_reset, self.device, "real")).reshape(n, 1, 1)
            state = {}
            for k, _ in self._state_keys():
                if requests.policy_state is None:
                    state[k] = torch.zeros((n, L, W), dtype=torch.float32, device=self.device)
                else:
                    state[k] = keep * fold(to_device_leaf(requests.policy_state[k], self.device, "real")).reshape(n, L, W)
        is_eval = np.asarray(requests.is_evaluation)
        if self._shared and is_eval.size == bs:  # one flag per environment: every agent of it evaluates (or not)
            is_eval = np.broadcast_to(is_eval.reshape(bs, 1), (bs, self._n_agents))
        action, logp, value, refs = self._rollout_rows(obs, n, is_eval, state)
        if self._use_popart and self.denormalize_value_during_rollout:
            value = self.denormalize_value(value)
        unfold = (lambda t: t.reshape(bs, self._n_agents, *t.shape[1:])) if self._shared else (lambda t: t)
        new_state = None
        if L:
            new_state = SMACPolicyState(*(unfold(self._net.last_state[tag].permute(1, 0, 2)).cpu().numpy()
                                          for _, tag in self._state_keys()))
        return policy_api.RolloutResult(action=SMACAction(unfold(action).cpu().numpy()),
                                        analyzed_result=PPORolloutAnalyzedResult(
                                            log_probs=unfold(logp).cpu().numpy(), value=unfold(value).cpu().numpy(),
                                            obs_ref=None if refs is None else unfold(refs.reshape(n, 1))),
                                        policy_state=new_state)


policy_api.register("smac_rnn", SMACPolicy)
