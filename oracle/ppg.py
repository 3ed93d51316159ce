"""Oracle: the auxiliary phase of Phasic Policy Gradient on PyTorch-CPU.

Restates, from ``legacy/algorithm/ppo/phasic_policy_gradient.py``:

* ``_policy_distance``   ``:140-144``  sum(KL(old || new) * undone) / sum(undone),  undone = 1 - done[..., 0]
* ``_paper_value_loss``  ``:146-147``  1/2 * sum(mse(value, target) * (1 - done)) / sum(1 - done)
* ``_compute_aux_loss``  ``:262-280``  aux_value_loss + beta_clone * sum_heads policy_distance + value_head_weight * value_head_loss,
  with ``done`` = the cache entry's ``info_mask``
* the auxiliary phase's inner loop  ``:232-243``  analyze -> loss -> zero_grad / backward / clip_grad_norm_ / aux optimiser step
* the distributions a cache entry keeps  ``:213-221``  (analysis without gradient on entering the phase; PopArt: the stored
  targets are normalised with the statistics of that moment)

The reference's ``MultiAgentPPG.step`` cannot run as shipped (``:169`` calls a method that does not exist, ``:174`` passes four
arguments to a three-argument ``_compute_loss``, and ``SampleBatch`` drops the ``value`` keyword of ``:193-201``): the PIECES above
do run and are what ``tests/golden/ppg.npz`` pins; the phase-1 glue around them has no reference behaviour to pin.

TEST INFRASTRUCTURE ONLY (see package doc).
"""
from typing import Dict, List, Optional

import numpy as np
import torch

from oracle.net import OracleActorCritic


def policy_distance(old_logq: torch.Tensor, new_logq: torch.Tensor, done: torch.Tensor) -> torch.Tensor:
    """:140-144 with torch.distributions.kl_divergence(Categorical, Categorical) written out (kl.py: p (log p - log q), +inf
    where q == 0, then 0 where p == 0)."""
    undone = 1.0 - done[..., 0]
    p, q = old_logq.exp(), new_logq.exp()
    t = p * (old_logq - new_logq)
    t = torch.where(q == 0, torch.full_like(t, float("inf")), t)
    t = torch.where(p == 0, torch.zeros_like(t), t)
    return (t.sum(-1) * undone).sum() / undone.sum()


def paper_value_loss(value, target, done):
    """:146-147."""
    return 0.5 * (((value - target)**2) * (1.0 - done)).sum() / (1.0 - done).sum()


def aux_loss(old_dists: List[torch.Tensor], new_dists: List[torch.Tensor], aux_value, pred_value, target, done, beta_clone=1.0,
             value_head_weight=1.0):
    """:262-280.  Returns (loss, dict of the three terms)."""
    pd = torch.stack([policy_distance(o, n, done) for o, n in zip(old_dists, new_dists)]).sum()
    av = paper_value_loss(aux_value, target, done)
    vh = paper_value_loss(pred_value, target, done)
    return av + beta_clone * pd + value_head_weight * vh, dict(auxiliary_value_loss=av, value_head_loss=vh, policy_distance=pd)


class OraclePPGAux:
    """The auxiliary phase over one cache entry (:205-243): ``enter`` keeps the current distributions (and normalises the stored
    targets under PopArt), ``epoch`` is one pass of the inner loop."""

    def __init__(self, net: OracleActorCritic, beta_clone=1.0, aux_value_head_weight=1.0, max_grad_norm=None, popart=False,
                 ppg_optimizer_config: Optional[dict] = None):
        self.net, self.beta, self.vhw, self.max_grad_norm, self.popart = net, beta_clone, aux_value_head_weight, max_grad_norm, popart
        self.optimizer = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], **(ppg_optimizer_config or {}))

    def _analyze(self, entry: Dict[str, np.ndarray]):
        f32 = lambda a: torch.from_numpy(np.asarray(a)).to(self.net.dtype)
        obs = {k[4:]: f32(v) for k, v in entry.items() if k.startswith("obs.")}
        names = ["policy_state.hx"] if self.net.shared else ["policy_state.actor_hx", "policy_state.critic_hx"]
        ps = [f32(entry[n]) for n in names] if self.net.num_rnn_layers else None
        return self.net.analyze_aux(obs, f32(entry["on_reset"]), ps)

    def enter(self, entry):
        with torch.no_grad():
            dists, _, _ = self._analyze(entry)
        self.old = [d.detach().clone() for d in dists]
        self.target = torch.from_numpy(np.asarray(entry["value"])).to(self.net.dtype)
        if self.popart:  # :222-225
            self.target = self.net.normalize_value(self.target)
        self.done = torch.from_numpy(np.asarray(entry["info_mask"])).to(self.net.dtype)

    def epoch(self, entry):
        dists, aux, pred = self._analyze(entry)
        loss, terms = aux_loss(self.old, dists, aux, pred, self.target, self.done, self.beta, self.vhw)
        self.optimizer.zero_grad()
        loss.backward()
        params = [p for p in self.net.parameters() if p.requires_grad]
        gn = torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm) if self.max_grad_norm is not None else None
        self.optimizer.step()
        out = {k: float(v.detach()) for k, v in terms.items()}
        out["loss"] = float(loss.detach())
        if gn is not None:
            out["grad_norm"] = float(gn)
        return out


    # ---- data parallel (MultiAgentPPG inherits MultiAgentPPO.distributed, :108; the policy is DistributedDataParallel): every rank
    # keeps the distributions of ITS entry, an epoch's loss is the rank's own masked means, the gradients are averaged over ranks
    def enter_dp(self, entries):
        self._ranks = []
        for e in entries:
            self.enter(e)
            self._ranks.append((self.old, self.target, self.done))

    def epoch_dp(self, entries):
        params = [p for p in self.net.parameters() if p.requires_grad]
        total, outs = [torch.zeros_like(p) for p in params], []
        for e, (old, target, done) in zip(entries, self._ranks):
            dists, aux, pred = self._analyze(e)
            loss, terms = aux_loss(old, dists, aux, pred, target, done, self.beta, self.vhw)
            for t, g in zip(total, torch.autograd.grad(loss, params, allow_unused=True)):
                if g is not None:
                    t += g
            outs.append({k: float(v.detach()) for k, v in terms.items()})
        self.optimizer.zero_grad()
        for p, t in zip(params, total):
            p.grad = t / len(entries)
        gn = torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm) if self.max_grad_norm is not None else None
        self.optimizer.step()
        return outs, (None if gn is None else float(gn))
