"""Gradients of a football-shaped step (full 676 M-parameter net, few rows) on the wide-dense pre-split path and on the
layer-by-layer kernels, both against the float64 CPU oracle's backward pass.  FB_T x FB_B rows (default 20 x 26);
SRL_H2_DENSE_MIN_ROWS must be <= that many rows for the pre-split path to engage."""
import math
import os
import sys
import time

os.environ.setdefault("SRL_H2_DENSE_MIN_ROWS", "256")
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _cheap_orthogonal(t, gain=1.0):
    with torch.no_grad():
        return t.normal_(0.0, gain / math.sqrt(t.shape[1] if t.dim() > 1 else t.numel()))


torch.nn.init.orthogonal_ = _cheap_orthogonal
import srl_amd
from oracle.net import OracleActorCritic
from oracle.trainer import OracleMappo
from srl_amd import hip
from srl_amd.algorithm.game_policies import FootballSMMPolicy
from srl_amd.algorithm.hipnet import HipNet
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic

srl_amd.register_all()
T, B, H = int(os.environ.get("FB_T", 20)), int(os.environ.get("FB_B", 26)), 128
TR = dict(popart=True, clip_value=True, value_loss="huber", value_loss_config=dict(delta=10.0), max_grad_norm=10.0,
          optimizer_config=dict(lr=5e-4, eps=1e-5))
arr = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec={"obs": ((4, 96, 72), "u8")}, action_dims=19, p_done=1 / 400,
                                   policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
out = {}
for dense in (True, False):
    HipNet.H2_DENSE = dense
    tr = trainer_api.make(config.Trainer("mappo", args=TR), config.Policy("football-smm-separate", args=dict(rnn_type="lstm", seed=1)))
    net = tr.policy.net
    if dense:
        sd = {k: v.numpy() for k, v in tr.policy.get_checkpoint()["state_dict"].items()}
    hip.dispatch_tiles(reset=True)
    res = tr.step(synthetic.to_sample_batch(arr))
    torch.cuda.synchronize()
    print(dense, {k: round(float(v), 6) for k, v in res.stats.items() if k in ("policy_loss", "value_loss", "entropy", "grad_norm")},
          sorted(k for k in hip.dispatch_tiles(reset=True) if k.startswith("h2:")), flush=True)
    # (the oracle's .grad is what clip_grad_norm_ left; the device clips inside its optimiser kernel)
    clip = min(1.0, TR["max_grad_norm"] / float(res.stats["grad_norm"]))
    out[dense] = {k: v.clone() * clip for k, v in net.flat_to_reference(net.grad.detach().cpu()).items()}
    del tr, net
    torch.cuda.empty_cache()
t0 = time.perf_counter()
pargs = dict(FootballSMMPolicy.defaults, rnn_type="lstm", seed=1,
             cnn_layers=dict(obs=[(4, 5, 1, 0, "zeros"), (8, 3, 1, 0, "zeros"), (4, 3, 1, 0, "zeros")]))
onet = OracleActorCritic(**pargs, dtype=torch.float64)
onet.load_state_dict(sd)
ostats, _ = OracleMappo(onet, **TR).step(arr)
print(f"float64 oracle step: {time.perf_counter() - t0:.1f} s", {k: round(float(ostats[k]), 6) for k in ("policy_loss", "value_loss", "entropy", "grad_norm")})
print("columns: max |g64|; max error / max |g64| of the pre-split path; of the layer-by-layer kernels")
for k, p in onet.params.items():
    g64 = p.grad.double()
    sc = float(g64.abs().max())
    e1, e0 = float((out[True][k].double() - g64).abs().max()), float((out[False][k].double() - g64).abs().max())
    flag = "  <<<<" if max(e1, e0) > 1e-3 * max(sc, 1e-30) else ""
    print(f"{k:62s} {sc:10.3e}  {e1 / max(sc, 1e-30):9.2e}  {e0 / max(sc, 1e-30):9.2e}{flag}")
