"""After one 256-row NatureCNN step: is the FC weight gradient in net.grad equal to dz^T x of the buffers the backward left
in the workspace?  usage: python3 scripts/fc_bwd_probe.py [f32]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "f32":
    os.environ["SRL_MFMA"] = "f32"
    os.environ["SRL_OBS_BF16"] = "0"
import srl_amd
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_trainer import ATARI_TRAINER, CNN_POLICY

srl_amd.register_all()
trainer = trainer_api.make(config.Trainer("mappo", args=dict(ATARI_TRAINER)), config.Policy("actor-critic", args=CNN_POLICY))
arrays = synthetic.make_sample_arrays(seed=70, T=16, B=16, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.05)
trainer.step(synthetic.to_sample_batch(arrays))
net = trainer.policy.net
n = 256
P = "a:obs_modules_dict.obs.1._Convolution__model."
print(sorted(k for k in net.ws._bufs if "7." in k or "model.4" in k))
dz = net.ws._bufs[P + "7.2.dx"][:n * 512].view(n, 512).double()
x = net.ws._bufs[P + "4.y"][:n * 3136].view(n, 3136).double()
info = net.spec.params["obs_modules_dict.obs.1._Convolution__model.7.0.weight"]
gw = net.grad[info.offset:info.offset + info.numel].view(512, 3136).double()
ref = dz.t() @ x
e = gw - ref
print("FC weight grad vs dz^T x:", float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()), float(e.abs().max() / ref.pow(2).mean().sqrt()))
ib = net.spec.params["obs_modules_dict.obs.1._Convolution__model.7.0.bias"]
gb = net.grad[ib.offset:ib.offset + ib.numel].double()
eb = gb - dz.sum(0)
print("FC bias grad vs colsum(dz):", float(eb.abs().max()), "scale", float(dz.sum(0).abs().mean()))
print("rows of dz that are entirely zero:", int((dz.abs().sum(1) == 0).sum()), " mask rows:", int((1 - arrays['on_reset'][1:17]).sum()))
tag = "f32" if len(sys.argv) > 1 and sys.argv[1] == "f32" else "bf16x3"
os.makedirs("gpurun_out/probe", exist_ok=True)
dump = {k: v.detach().cpu().numpy().copy() for k, v in net.ws._bufs.items()
        if k in ("d_logits", "d_value", "d_logp", "d_entropy", "new_logp", "entropy", "logits", "value", "a:actor_head.dx", P + "7.2.dx", P + "7.0.dx", P + "7.2.y")}
print({k: v.shape for k, v in dump.items()})
np.savez(f"gpurun_out/probe/bufs_{tag}.npz", **dump)
