"""Oracle: GAE / V-trace reverse scan and the actor-side trajectory GAE (numpy, float64).

Restates ``legacy/algorithm/modules/gae.py`` of the reference:
``gae_trace`` (:8-97) and ``TrajGAE.process`` (:100-139).  TEST INFRASTRUCTURE ONLY (see package doc).
"""
import numpy as np


def check_invariants(reward, value, truncated, done, on_reset):
    """The five debug assertions of the reference (gae.py:69-77) minus the two that need delta."""
    assert (truncated * done == 0).all()
    assert ((truncated + done)[:-1] == on_reset[1:]).all()
    assert (reward * on_reset[1:] == 0).all()


def gae_trace(reward, value, truncated, done, on_reset, gamma, lmbda, vtrace=False, imp_ratio=None,
              rho=1.0, c=1.0):
    """adv[t] = delta[t] + m[t] * adv[t+1], adv[T] = 0, in float64, returned as float32.

    reward [T, B, Nc]; value [T+1, B, Nc]; truncated / done / on_reset [T+1, B, 1];
    gamma / lmbda python floats or arrays [T, B, 1].  (reference gae.py:46-97)
    """
    f64 = lambda x: np.asarray(x, dtype=np.float64)
    reward, value, truncated, done, on_reset = map(f64, (reward, value, truncated, done, on_reset))
    gamma = f64(gamma) if not isinstance(gamma, float) else gamma
    lmbda = f64(lmbda) if not isinstance(lmbda, float) else lmbda
    not_reset_next = 1.0 - on_reset[1:]
    delta = reward + gamma * value[1:] * not_reset_next - value[:-1]  # :63
    carry = gamma * lmbda * not_reset_next * (1.0 - truncated[1:])  # :87
    if vtrace:
        ratio = f64(imp_ratio)
        delta = delta * np.minimum(ratio, rho)  # :64-65
        carry = carry * np.minimum(ratio, c)  # :88-89
    adv = np.zeros_like(reward)
    running = np.zeros_like(reward[0])
    for t in range(reward.shape[0] - 1, -1, -1):  # :91-95
        running = delta[t] + carry[t] * running
        adv[t] = running
    return adv.astype(np.float32)


def adv_and_value_target(reward, value, truncated, done, on_reset, gamma, lmbda, **kw):
    """``_compute_adv_and_value_target`` without PopArt (reference mappo.py:118-144).

    reward has Tb rows here (the last is dropped, :131); returns (adv, ret) with Tb-1 rows.
    The masking and the final addition happen in float32 like the reference's torch ops.
    """
    masked_value = (np.asarray(value, np.float32) * (1 - np.asarray(done, np.float32))).astype(np.float32)
    adv = gae_trace(np.asarray(reward, np.float32)[:-1], masked_value, truncated, done, on_reset, gamma, lmbda,
                    **kw)
    return adv, (adv + masked_value[:-1]).astype(np.float32)


def traj_gae(rewards, values, truncated_last, value_last, gamma, lmbda):
    """Actor-side per-episode GAE (reference gae.py:109-139).

    rewards / values: lists over the L-1 non-final steps; ``value_last`` is the value estimate of the
    final observation or None when the final step carries no analyzed result; it is used only if the
    final step is ``truncated`` (:121-127).  Returns (adv list, ret list) for steps 0..L-2.
    """
    L1 = len(rewards)
    adv = [None] * L1
    ret = [None] * L1
    gae = np.zeros_like(np.asarray(rewards[0]))
    for t in range(L1 - 1, -1, -1):
        if t == L1 - 1:
            boot = 0 if value_last is None else np.asarray(value_last) * np.asarray(truncated_last)
        else:
            boot = values[t + 1]
        delta = rewards[t] + gamma * boot - values[t]
        gae = gamma * lmbda * gae + delta
        adv[t] = gae
        ret[t] = gae + values[t]
    return adv, ret
