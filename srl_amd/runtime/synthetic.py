"""Synthetic ``[Tb, B]`` samples with the reference's data invariants.

There is no ALE / SC2 on the GPU box, so benchmarks and parity tests run on synthetic rollouts
that satisfy exactly the invariants the reference's ``gae_trace`` asserts in debug mode
(``legacy/algorithm/modules/gae.py:69-77``):

* ``truncated * done == 0``
* ``on_reset[0] == 0`` and ``on_reset[t+1] == done[t] | truncated[t]``
* ``reward[t] == 0`` where ``on_reset[t+1]`` (the final, reset-triggering step carries no reward)
* ``value == 0`` where ``done``

All leaves carry the wire dtypes the actor worker produces (flags ``uint8``, reward / value /
log-prob ``float32``, action ``int32``, ``policy_version_steps`` ``int64``) and a trailing singleton
dimension, time-major ``[Tb, B, ...]``.
"""
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np

ObsSpec = Dict[str, Tuple[Tuple[int, ...], str]]  # name -> (shape, "f32" | "u8")


def make_flags(rng: np.random.Generator, Tb: int, B: int, p_done: float, p_trunc: Optional[float] = None):
    """done / truncated / on_reset ``[Tb, B, 1]`` uint8 obeying the invariants above."""
    p_trunc = p_done / 4 if p_trunc is None else p_trunc
    done = (rng.random((Tb, B, 1)) < p_done)
    truncated = (rng.random((Tb, B, 1)) < p_trunc) & ~done
    on_reset = np.zeros((Tb, B, 1), dtype=bool)
    on_reset[1:] = done[:-1] | truncated[:-1]
    # an observation cannot be both the first of an episode and terminal in these synthetic rollouts
    # (keeps every episode at least two steps long, as real environments do)
    done &= ~on_reset
    truncated &= ~on_reset
    on_reset[1:] = done[:-1] | truncated[:-1]
    return done.astype(np.uint8), truncated.astype(np.uint8), on_reset.astype(np.uint8)


def make_sample_arrays(seed: int,
                       T: int,
                       B: int,
                       obs_spec: ObsSpec,
                       action_dims: Union[int, Sequence[int]],
                       p_done: float = 0.05,
                       bootstrap_steps: int = 1,
                       value_dim: int = 1,
                       available_action: bool = False,
                       p_trunc: Optional[float] = None,
                       policy_state: Optional[Dict[str, Tuple[int, int]]] = None,
                       continuous_action: bool = False,
                       flags=None) -> Dict[str, np.ndarray]:
    """Flat ``{dotted.key: array}`` dict of one synthetic sample; wrap with ``to_sample_batch``.

    ``policy_state``: ``{"hx": (layers, hidden)}`` (or ``actor_hx`` / ``critic_hx``) adds stored recurrent states
    ``[Tb, B, layers, hidden]`` (drawn from their own stream, so the other leaves do not depend on it)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Tb = T + bootstrap_steps
    done, truncated, on_reset = make_flags(rng, Tb, B, p_done, p_trunc) if flags is None else flags
    value = (rng.standard_normal((Tb, B, value_dim)) * (1 - done)).astype(np.float32)
    reward = rng.standard_normal((Tb, B, value_dim)).astype(np.float32)
    reward[:-1] *= (1 - on_reset[1:])
    dims = [action_dims] if isinstance(action_dims, (int, np.integer)) else list(action_dims)
    action = np.stack([rng.integers(0, d, size=(Tb, B)) for d in dims], axis=-1).astype(np.int32)
    log_probs = (sum(np.log(1.0 / d) for d in dims) + 0.01 * rng.standard_normal((Tb, B, 1))).astype(np.float32)
    out = {}
    for name, (shape, kind) in obs_spec.items():
        if kind == "u8":
            out[f"obs.{name}"] = rng.integers(0, 256, size=(Tb, B, *shape), dtype=np.uint8)
        else:
            out[f"obs.{name}"] = rng.standard_normal((Tb, B, *shape)).astype(np.float32)
    if available_action:
        assert len(dims) == 1
        avail = (rng.random((Tb, B, dims[0])) < 0.7)
        avail[np.arange(Tb)[:, None], np.arange(B)[None, :], action[..., 0]] = True  # taken action is legal
        out["obs.available_action"] = avail.astype(np.uint8)
    out.update({
        "on_reset": on_reset,
        "done": done,
        "truncated": truncated,
        "action.x": action,
        "reward": reward,
        "analyzed_result.log_probs": log_probs,
        "analyzed_result.value": value,
        "policy_version_steps": np.zeros((Tb, B, 1), dtype=np.int64),
        "info_mask": np.zeros((Tb, B, 1), dtype=np.uint8),
    })
    if continuous_action:  # float32 actions [Tb, B, A] and log-probs of a unit Gaussian around them
        crng = np.random.Generator(np.random.PCG64(seed + 104729))
        a = crng.standard_normal((Tb, B, int(sum(dims)))).astype(np.float32)
        out["action.x"] = a
        out["analyzed_result.log_probs"] = (-0.5 * a.shape[-1] * np.log(2 * np.pi) - 0.5 * (a**2).sum(-1, keepdims=True) +
                                            0.01 * crng.standard_normal((Tb, B, 1))).astype(np.float32)
    if policy_state:
        srng = np.random.Generator(np.random.PCG64(seed + 7919))
        for name, (layers, hidden) in policy_state.items():
            out[f"policy_state.{name}"] = (0.5 * srng.standard_normal((Tb, B, layers, hidden))).astype(np.float32)
    return out


def make_multiagent_arrays(seed: int, T: int, B: int, agents: int, obs_spec: ObsSpec, action_dim: int, p_done: float = 0.05,
                           p_dead: float = 0.15, policy_state: Optional[Dict[str, Tuple[int, int]]] = None,
                           bootstrap_steps: int = 1) -> Dict[str, np.ndarray]:
    """One sample of a *shared* multi-agent environment: every leaf ``[Tb, B, agents, ...]`` (smac_env.py:229-241).
    Episode flags are per environment and identical for its agents (``dones[:] = dones.all()``); observations,
    actions, rewards and values are per agent; ``obs.is_alive`` marks dead agents, ``obs.available_action`` masks
    actions (the taken one is always legal)."""
    rng = np.random.Generator(np.random.PCG64(seed + 15485863))
    Tb = T + bootstrap_steps
    env_flags = make_flags(rng, Tb, B, p_done)
    flags = tuple(np.repeat(f, agents, axis=1) for f in env_flags)
    flat = make_sample_arrays(seed, T, B * agents, obs_spec, action_dim, p_done, bootstrap_steps, available_action=True,
                              policy_state=policy_state, flags=flags)
    flat["obs.is_alive"] = (rng.random((Tb, B * agents, 1)) >= p_dead).astype(np.uint8)
    return {k: v.reshape(Tb, B, agents, *v.shape[2:]) for k, v in flat.items()}


def to_sample_batch(arrays: Dict[str, np.ndarray]):
    """Wrap the flat dict into this package's ``SampleBatch`` (field classes as the runtime uses)."""
    from srl_amd.api.env_utils import DiscreteAction
    from srl_amd.api.trainer import SampleBatch
    from srl_amd.algorithm.ppo_types import PPORolloutAnalyzedResult
    from srl_amd.namedarray import NamedArray
    obs = NamedArray(**{k[4:]: v for k, v in arrays.items() if k.startswith("obs.")})
    ps = {k[len("policy_state."):]: v for k, v in arrays.items() if k.startswith("policy_state.")}
    return SampleBatch(obs=obs,
                       policy_state=NamedArray(**ps) if ps else None,
                       on_reset=arrays["on_reset"],
                       done=arrays["done"],
                       truncated=arrays["truncated"],
                       action=DiscreteAction(arrays["action.x"]),
                       reward=arrays["reward"],
                       analyzed_result=PPORolloutAnalyzedResult(log_probs=arrays["analyzed_result.log_probs"],
                                                                value=arrays["analyzed_result.value"],
                                                                obs_ref=arrays.get("analyzed_result.obs_ref")),
                       policy_version_steps=arrays["policy_version_steps"],
                       info_mask=arrays["info_mask"])


CARTPOLE_OBS: ObsSpec = {"obs": ((4,), "f32")}
ATARI_OBS: ObsSpec = {"obs": ((4, 84, 84), "u8")}
