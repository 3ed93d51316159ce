"""Football-sized step on the wide-dense pre-split path (HipNet.H2_DENSE) against the layer-by-layer kernels: per-tensor gradient
differences and the step's statistics.  FB_B / FB_T as in football_bench.py."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _cheap_orthogonal(t, gain=1.0):
    with torch.no_grad():
        return t.normal_(0.0, gain / math.sqrt(t.shape[1] if t.dim() > 1 else t.numel()))


torch.nn.init.orthogonal_ = _cheap_orthogonal
import srl_amd
from srl_amd.algorithm.hipnet import HipNet
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic

srl_amd.register_all()
T, B, H = int(os.environ.get("FB_T", 200)), int(os.environ.get("FB_B", 256)), 128
arr = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec={}, action_dims=19, p_done=1 / 400,
                                   policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
dev = {k: torch.from_numpy(v).to("cuda:0") for k, v in arr.items()}
gen = torch.Generator(device="cuda").manual_seed(0)
dev["obs.obs"] = torch.randint(0, 256, (T + 1, B, 4, 96, 72), dtype=torch.uint8, device="cuda", generator=gen)
out = {}
for run, dense in enumerate((True, False, False)):
    HipNet.H2_DENSE = dense
    torch.manual_seed(1)
    tr = trainer_api.make(config.Trainer("mappo", args=dict(popart=True, clip_value=True, value_loss="huber",
                                                            value_loss_config=dict(delta=10.0), max_grad_norm=10.0,
                                                            optimizer_config=dict(lr=5e-4, eps=1e-5))),
                          config.Policy("football-smm-separate", args=dict(rnn_type="lstm", seed=1)))
    net = tr.policy.net
    flat0 = net.flat.clone()
    res = tr.step(synthetic.to_sample_batch(dict(dev)))
    torch.cuda.synchronize()
    out[run] = (res.stats, {k: v.clone() for k, v in net.flat_to_reference(net.grad.detach().cpu()).items()}, flat0.cpu())
    print(dense, {k: round(float(v), 6) for k, v in res.stats.items() if k in ("policy_loss", "value_loss", "entropy", "grad_norm")}, flush=True)
    del tr, net
    torch.cuda.empty_cache()
print("same initial parameters:", bool(torch.equal(out[0][2], out[1][2])))
print("columns: max |g|; pre-split path vs layer-by-layer; layer-by-layer run twice (the noise floor)")
for k, g1 in out[0][1].items():
    g0, g2 = out[1][1][k], out[2][1][k]
    sc = float(g0.abs().max())
    d, dn = float((g1 - g0).abs().max()), float((g2 - g0).abs().max())
    flag = "  <<<<" if d > 1e-3 * max(sc, 1e-30) else ""
    print(f"{k:62s} {sc:10.3e}  {d / max(sc, 1e-30):9.2e}  {dn / max(sc, 1e-30):9.2e}{flag}")
