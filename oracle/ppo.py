"""Oracle: masked advantage normalisation and the PPO loss (restating the reference op for op).

``masked_normalization``  <- ``legacy/algorithm/modules/utils.py:10-67``   (numpy float64)
``value_loss_elementwise`` <- ``legacy/algorithm/modules/utils.py:228-265`` (torch CPU float32)
``ppo_loss``              <- ``legacy/algorithm/ppo/mappo.py:146-217``     (torch CPU float32, autograd)
TEST INFRASTRUCTURE ONLY (see package doc).
"""
import numpy as np
import torch


def masked_stats(x, mask=None):
    """(n, sum, sum of squares) in float64 over all dims of x*mask (utils.py:38-57)."""
    x = np.asarray(x, dtype=np.float64)
    if mask is None:
        return float(x.size), float(x.sum()), float(np.square(x).sum())
    mask = np.asarray(mask, dtype=np.float64)
    xm = x * mask
    return float(mask.sum()), float(xm.sum()), float(np.square(xm).sum())


def masked_normalization(x, mask=None, unbiased=False, eps=1e-5, stats=None):
    """((x*mask) - mean) / (sqrt(var) + eps) in float64, returned float32 (utils.py:38-67).

    ``stats`` = (n, s, q) overrides the locally computed sums (the all-reduced values when data parallel,
    utils.py:58-61).  Note masked-out entries come out as -mean/(std+eps), not 0, like the reference.
    """
    xm = np.asarray(x, dtype=np.float64)
    if mask is not None:
        xm = xm * np.asarray(mask, dtype=np.float64)
    n, s, q = masked_stats(x, mask) if stats is None else stats
    mean = s / n
    var = q / n - mean**2
    if unbiased:
        var *= n / (n - 1)
    return ((xm - mean) / (np.sqrt(var) + eps)).astype(np.float32)


def value_loss_elementwise(name, value, target, old_value=None, clip_value=False, value_eps_clip=0.2, delta=1.0,
                           beta=1.0):
    """Per-sample value loss: mse | huber(delta) | smoothl1(beta), optionally PPO-clipped (utils.py:228-265)."""

    def base(v, t):
        d = v - t
        if name == "mse":
            return d * d
        a = d.abs()
        if name == "huber":
            return torch.where(a <= delta, 0.5 * d * d, delta * (a - 0.5 * delta))
        if name == "smoothl1":
            return torch.where(a < beta, 0.5 * d * d / beta, a - 0.5 * beta)
        raise ValueError(name)

    plain = base(value, target)
    if not clip_value:
        return plain
    clipped = old_value + (value - old_value).clamp(-value_eps_clip, value_eps_clip)
    return torch.max(plain, base(clipped, target))


def ppo_loss(new_lp, old_lp, value, old_value, adv, ret, entropy, mask, *, eps_clip=0.2, dual_clip=True,
             c_clip=3.0, value_loss="mse", value_loss_config=None, clip_value=False, value_eps_clip=None,
             value_loss_weight=0.5, entropy_bonus_weight=0.01, norm_stats=None):
    """Scalar PPO loss + the per-term statistics the reference logs (mappo.py:146-217).

    All inputs torch float32 tensors of identical shape [T, B, 1] (mask 0/1).  ``new_lp``, ``value`` and
    ``entropy`` may require grad.  Returns (loss, dict of python floats).
    """
    cfg = dict(value_loss_config or {})
    value_eps_clip = eps_clip if value_eps_clip is None else value_eps_clip
    msum = mask.sum()
    masked_mean = lambda t: (t * mask).sum() / msum

    vl = value_loss_elementwise(value_loss, value, ret, old_value, clip_value, value_eps_clip,
                                delta=cfg.get("delta", 1.0), beta=cfg.get("beta", 1.0))
    v_loss = masked_mean(vl)  # :184

    ratio = (new_lp - old_lp).exp()  # :157-158
    norm_adv = torch.from_numpy(
        masked_normalization(adv.detach().numpy(), mask.numpy(), stats=norm_stats))  # :187
    s1 = ratio * norm_adv
    s2 = ratio.clamp(1 - eps_clip, 1 + eps_clip) * norm_adv
    if dual_clip:
        s3 = -torch.sign(norm_adv) * c_clip * norm_adv
        p_elem = -torch.max(torch.min(s1, s2), s3)  # :191-193
    else:
        p_elem = -torch.min(s1, s2)
    p_loss = masked_mean(p_elem)  # :197
    e_loss = -masked_mean(entropy)  # :199
    loss = p_loss + value_loss_weight * v_loss + entropy_bonus_weight * e_loss  # :202

    sel = mask.bool()
    pick = lambda t: torch.masked_select(t.detach(), sel).mean().item()
    stats = dict(advantage=pick(adv), entropy=(-e_loss).item(), policy_loss=p_loss.item(),
                 value_loss=v_loss.item(), importance_weight=pick(ratio),
                 clip_ratio=pick((s2 < s1).float()), value_targets=pick(ret))
    return loss, stats
