"""Determinism of the first update with one and with two row-chunk pipelines (fresh trainers, same device sample)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import srl_amd
from bench import POLICY, TRAINER, device_sample
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic
srl_amd.register_all()
sample = device_sample(11, 128, 384, "cuda:0")
for pipes in (1, 2, 2, 2, 2):
    for side in ("0",):
        os.environ["SRL_WGRAD_STREAM"] = side
        tr = trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, chunk_rows=16384, pipelines=pipes)), config.Policy("actor-critic", args=POLICY))
        smp = synthetic.to_sample_batch({k: v for k, v in [("obs.obs", sample.obs.obs), ("on_reset", sample.on_reset), ("done", sample.done), ("truncated", sample.truncated), ("action.x", sample.action.x), ("reward", sample.reward), ("analyzed_result.log_probs", sample.analyzed_result.log_probs), ("analyzed_result.value", sample.analyzed_result.value), ("policy_version_steps", sample.policy_version_steps), ("info_mask", sample.info_mask)]})
        r = tr.step(smp)
        g = tr.policy.net.grad
        print(f"pipelines={pipes} side={side} policy_loss {r.stats['policy_loss']:.10e} value_loss {r.stats['value_loss']:.10e} grad_norm {r.stats['grad_norm']:.8f} gradsum {float(g.double().sum()):.10e}")
        net = tr.policy.net
        ex = net  # three chunks: the last one (2) ran on the policy's own executor in both modes
        cur = {k: ex.ws._bufs[k].clone() for k in ("logits", "value", "new_logp", "entropy", "d_logp", "d_value", "d_logits", "a:obs_modules_dict.obs.1._Convolution__model.0.y", "a:obs_modules_dict.obs.1._Convolution__model.2.y", "a:obs_modules_dict.obs.1._Convolution__model.4.y", "a:obs_modules_dict.obs.1._Convolution__model.7.0.y", "a:obs_modules_dict.obs.1._Convolution__model.7.2.y")}
        cur["act_absmax"] = ex.ws._bufs["act_absmax"][:4].clone()
        cur["wamax"] = net._wamax[:4].clone()
        cur["terms"] = net.ws._bufs["mappo.out"][:44].clone()
        if "ref" not in globals():
            ref = cur
        else:
            print("   terms diff per chunk (policy, value):", [(float(cur["terms"][11 * c] - ref["terms"][11 * c]), float(cur["terms"][11 * c + 1] - ref["terms"][11 * c + 1])) for c in range(3)])
            dl = (cur["logits"][:16384 * 6].view(16384, 6) != ref["logits"][:16384 * 6].view(16384, 6)).any(1).nonzero().flatten().tolist()
            dv = (cur["value"][:16384] != ref["value"][:16384]).nonzero().flatten().tolist()
            print("   rows with different logits:", len(dl), dl[:12], "...", dl[-6:], " value rows:", len(dv), dv[:12])
            print("   vs first run:", {k[-12:]: (float((cur[k] - ref[k]).abs().max()), int((cur[k] != ref[k]).sum())) for k in cur})
