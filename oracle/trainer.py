"""Oracle: one full ``MultiAgentPPO.step`` on PyTorch-CPU (restating reference ``legacy/algorithm/ppo/mappo.py``).

Follows ``mappo.py:68-116`` (hyper-parameters and defaults), ``:118-144`` (advantage / value target),
``:146-217`` (loss), ``:219-328`` (step driver: analysis, GAE, zero-padded adv/ret written back,
``loss_mask = 1 - on_reset[1+burn_in : 1+Tb-boot]``, backward, global-L2 clip or measure, optimiser step,
stats averaged over ``ppo_epochs``, ``policy.version`` += 1).  The reference's GPU prefetcher is a
pass-through here: every leaf is converted to float32 like ``api/trainer.py:217``.

This is also the CPU baseline timed by ``bench.py`` (``cpu_baseline.kind == "port"``).
TEST INFRASTRUCTURE ONLY (see package doc).  Samples are flat ``{dotted.key: numpy array}`` dicts.
"""
from typing import Dict

import numpy as np
import torch

from oracle import gae as ogae
from oracle import ppo as oppo
from oracle.net import OracleActorCritic


class OracleMappo:

    def __init__(self, net: OracleActorCritic, **kw):
        self.net = net
        g = kw.get
        self.discount_rate = g("discount_rate", 0.99)
        self.gae_lambda = g("gae_lambda", 0.97)
        self.eps_clip = g("eps_clip", 0.2)
        self.clip_value = g("clip_value", False)
        self.dual_clip = g("dual_clip", True)
        self.c_clip = g("c_clip", 3)
        self.burn_in_steps = g("burn_in_steps", 0)
        self.vtrace = g("vtrace", False)
        self.value_eps_clip = g("value_eps_clip", self.eps_clip)
        self.value_loss_weight = g("value_loss_weight", 0.5)
        self.entropy_bonus_weight = g("entropy_bonus_weight", 0.01)
        self.max_grad_norm = g("max_grad_norm")
        self.bootstrap_steps = g("bootstrap_steps", 1)
        self.ppo_epochs = g("ppo_epochs", 1)
        self.value_loss = g("value_loss", "mse")
        self.value_loss_config = g("value_loss_config", {})
        self.popart = g("popart", False)
        assert not g("normalize_old_value", False), "oracle: normalize_old_value not restated"
        opt_cls = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "rmsprop": torch.optim.RMSprop,
                   "sgd": torch.optim.SGD}[g("optimizer", "adam")]  # modules/utils.py:268-286
        self.optimizer = opt_cls([p for p in net.parameters() if p.requires_grad], **g("optimizer_config", {}))
        self.version = -1
        self.frames = 0

    def step(self, sample: Dict[str, np.ndarray]):
        f32 = lambda k: torch.from_numpy(np.asarray(sample[k])).to(self.net.dtype)  # float32 (api/trainer.py:217)
        on_reset, done, truncated = f32("on_reset"), f32("done"), f32("truncated")
        reward, old_value, old_lp = f32("reward"), f32("analyzed_result.value"), f32("analyzed_result.log_probs")
        action = torch.from_numpy(np.asarray(sample["action.x"])).to(self.net.dtype)
        obs = {k[4:]: f32(k) for k in sample if k.startswith("obs.")}
        pstate = None
        if self.net.num_rnn_layers:
            names = ["policy_state.hx"] if self.net.shared else ["policy_state.actor_hx", "policy_state.critic_hx"]
            pstate = [f32(n) for n in names]
        Tb = on_reset.shape[0]
        boot, burn = self.bootstrap_steps, self.burn_in_steps
        totals = {}
        out = {}
        for _ in range(self.ppo_epochs):
            # analysed rows (mappo.py:243-246): tail_len = bootstrap_steps, or 1 while V-trace still needs the importance ratio
            # of every rewarding step (the loss then takes the first Tb - boot of them, :159-163)
            keep = Tb - 1 if (self.vtrace and "adv" not in out) else Tb - boot
            lp, value, ent, _ = self.net.analyze({k: v[:keep] for k, v in obs.items()}, action[:keep], on_reset[:keep],
                                                 None if pstate is None else [s[:keep] for s in pstate], burn)
            if burn:  # the analysis covers rows [burn, keep): pad in front so that row indices stay the sample's
                padf = lambda x: torch.cat([torch.zeros((burn,) + tuple(x.shape[1:]), dtype=x.dtype), x], 0)
                lp, value, ent = padf(lp), padf(value), padf(ent)
            if "adv" not in out:  # computed in the first epoch only (mappo.py:247-257; matters with PopArt, whose
                # statistics move between epochs)
                trace_value = self.net.denormalize_value(old_value) if self.popart else old_value  # :120-124
                kw = {}
                if self.vtrace:  # :130-133: the analysed rows are the Tb - 1 rewarding rows
                    assert boot >= 1 and not burn, "oracle: V-trace needs a bootstrap row; with burn-in the reference's shapes clash"
                    assert boot == 1 or not self.net.num_rnn_layers, "oracle: V-trace with bootstrap_steps > 1 restated for feed-forward nets"
                    kw = dict(vtrace=True, imp_ratio=(lp - old_lp[:keep]).exp().detach().numpy())
                adv, ret = ogae.adv_and_value_target(reward.numpy(), trace_value.numpy(), truncated.numpy(), done.numpy(),
                                                     on_reset.numpy(), self.discount_rate, self.gae_lambda, **kw)
                pad = lambda x: np.concatenate([x, np.zeros_like(x[:1])], 0)  # mappo.py:254-256
                adv_p, ret_p = pad(adv), pad(ret)
                out["adv"], out["ret"] = adv_p, ret_p
            lo, hi = burn, Tb - boot
            mask = 1 - on_reset[1 + lo:1 + hi]  # mappo.py:260-261
            target = torch.from_numpy(ret_p[lo:hi])
            if self.popart:  # mappo.py:263-264 then :173-176
                self.net.update_popart(target, mask)
                denorm_target, target = target, self.net.normalize_value(target)
            loss, stats = oppo.ppo_loss(lp[lo:hi], old_lp[lo:hi], value[lo:hi], old_value[lo:hi],
                                        torch.from_numpy(adv_p[lo:hi]), target, ent[lo:hi],
                                        mask, eps_clip=self.eps_clip, dual_clip=self.dual_clip, c_clip=self.c_clip,
                                        value_loss=self.value_loss, value_loss_config=self.value_loss_config,
                                        clip_value=self.clip_value, value_eps_clip=self.value_eps_clip,
                                        value_loss_weight=self.value_loss_weight,
                                        entropy_bonus_weight=self.entropy_bonus_weight)
            self.optimizer.zero_grad(set_to_none=True)
            loss.backward()
            if self.popart:
                stats["denorm_value"] = torch.masked_select(denorm_target, mask.bool()).mean().item()
            params = [p for p in self.net.parameters() if p.requires_grad]
            if self.max_grad_norm is not None:
                gn = torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm)
            else:
                gn = torch.sqrt(sum(p.grad.norm()**2 for p in params if p.grad is not None))
            self.optimizer.step()
            stats["grad_norm"] = float(gn)
            stats["loss"] = float(loss.detach())
            stats["done"] = done[lo:hi].mean().item()
            stats["truncated"] = truncated[lo:hi].mean().item()
            for k, v in stats.items():
                totals[k] = totals.get(k, 0.0) + v
        stats = {k: v / self.ppo_epochs for k, v in totals.items()}
        self.version += 1
        self.frames += int(np.prod(on_reset[burn:Tb - boot].shape))
        stats["frames"] = self.frames
        return stats, out

    def step_dp(self, samples):
        """One DATA-PARALLEL step of ``len(samples)`` ranks, emulated on the CPU: what the reference does when every
        trainer rank holds its own columns of the batch.  Gradients are the mean over ranks of each rank's gradient
        (DistributedDataParallel, ``api/policy.py:219-238``), each rank's loss terms being masked means over its LOCAL
        steps (``mappo.py:184-199``); the advantage statistics (``modules/utils.py:58-61``) and the PopArt statistics
        (``:121-124``) are sums over all ranks.  Returns (list of per-rank stats, list of per-rank padded adv/ret)."""
        assert not self.vtrace and not self.burn_in_steps and not self.net.num_rnn_layers, "oracle DP: feed-forward PPO only"
        W = len(samples)
        f32 = lambda smp, k: torch.from_numpy(np.asarray(smp[k])).to(self.net.dtype)
        boot = self.bootstrap_steps
        ranks = []
        for smp in samples:
            r = dict(on_reset=f32(smp, "on_reset"), done=f32(smp, "done"), truncated=f32(smp, "truncated"),
                     reward=f32(smp, "reward"), old_value=f32(smp, "analyzed_result.value"),
                     old_lp=f32(smp, "analyzed_result.log_probs"),
                     action=torch.from_numpy(np.asarray(smp["action.x"])).to(self.net.dtype),
                     obs={k[4:]: f32(smp, k) for k in smp if k.startswith("obs.")})
            Tb = r["on_reset"].shape[0]
            r["keep"] = Tb - boot
            r["mask"] = 1 - r["on_reset"][1:1 + r["keep"]]
            ranks.append(r)
        totals = [dict() for _ in range(W)]
        for _ in range(self.ppo_epochs):
            for r in ranks:
                k = r["keep"]
                r["lp"], r["value"], r["ent"], _ = self.net.analyze({n: v[:k] for n, v in r["obs"].items()}, r["action"][:k],
                                                                    r["on_reset"][:k], None, 0)
                if "adv" not in r:
                    tv = self.net.denormalize_value(r["old_value"]) if self.popart else r["old_value"]
                    adv, ret = ogae.adv_and_value_target(r["reward"].numpy(), tv.numpy(), r["truncated"].numpy(),
                                                         r["done"].numpy(), r["on_reset"].numpy(), self.discount_rate,
                                                         self.gae_lambda)
                    pad = lambda x: np.concatenate([x, np.zeros_like(x[:1])], 0)
                    r["adv"], r["ret"] = pad(adv), pad(ret)
            # the three sums of the advantage normalisation, over all ranks
            parts = [oppo.masked_stats(r["adv"][:r["keep"]], r["mask"].numpy()) for r in ranks]
            gstats = tuple(sum(p[i] for p in parts) for i in range(3))
            targets = [torch.from_numpy(r["ret"][:r["keep"]]) for r in ranks]
            if self.popart:  # one update from the statistics of every rank's targets
                self.net.update_popart(torch.cat(targets, 1), torch.cat([r["mask"] for r in ranks], 1))
                denorm = targets
                targets = [self.net.normalize_value(t) for t in targets]
            self.optimizer.zero_grad(set_to_none=True)
            total = 0.0
            per_rank = []
            for r, tgt in zip(ranks, targets):
                k = r["keep"]
                loss, st = oppo.ppo_loss(r["lp"], r["old_lp"][:k], r["value"], r["old_value"][:k],
                                         torch.from_numpy(r["adv"][:k]), tgt, r["ent"], r["mask"], eps_clip=self.eps_clip,
                                         dual_clip=self.dual_clip, c_clip=self.c_clip, value_loss=self.value_loss,
                                         value_loss_config=self.value_loss_config, clip_value=self.clip_value,
                                         value_eps_clip=self.value_eps_clip, value_loss_weight=self.value_loss_weight,
                                         entropy_bonus_weight=self.entropy_bonus_weight, norm_stats=gstats)
                total = total + loss / W
                per_rank.append(st)
            total.backward()
            params = [p for p in self.net.parameters() if p.requires_grad]
            if self.max_grad_norm is not None:
                gn = torch.nn.utils.clip_grad_norm_(params, self.max_grad_norm)
            else:
                gn = torch.sqrt(sum(p.grad.norm()**2 for p in params if p.grad is not None))
            self.optimizer.step()
            for i, st in enumerate(per_rank):
                st["grad_norm"] = float(gn)
                if self.popart:
                    st["denorm_value"] = torch.masked_select(denorm[i], ranks[i]["mask"].bool()).mean().item()
                for kk, v in st.items():
                    totals[i][kk] = totals[i].get(kk, 0.0) + v
        self.version += 1
        stats = [{k: v / self.ppo_epochs for k, v in t.items()} for t in totals]
        return stats, [dict(adv=r["adv"], ret=r["ret"]) for r in ranks]
