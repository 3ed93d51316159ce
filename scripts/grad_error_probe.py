"""Gradient error of one NatureCNN step (256 rows, one chunk) against a FLOAT64 restatement, per parameter tensor and
relative to the tensor's rms gradient: the device path (bf16x3 kernels, or `f32`: the float32 MFMA kernels) next to the
float32 oracle's own error.  usage: python3 scripts/grad_error_probe.py [noise|pong] [f32]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
frames_kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
if len(sys.argv) > 2 and sys.argv[2] == "f32":
    os.environ["SRL_MFMA"] = "f32"
    os.environ["SRL_OBS_BF16"] = "0"
import srl_amd
from oracle.net import OracleActorCritic
from oracle.trainer import OracleMappo
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_trainer import ATARI_TRAINER, CNN_POLICY, _pong_frames

srl_amd.register_all()
trainer = trainer_api.make(config.Trainer("mappo", args=dict(ATARI_TRAINER)), config.Policy("actor-critic", args=CNN_POLICY))
init = {k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()}
arrays = synthetic.make_sample_arrays(seed=int(os.environ.get("PROBE_SEED", "70")), T=16, B=16, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.05)
if frames_kind == "pong":
    arrays["obs.obs"] = _pong_frames(np.random.default_rng(0), 17, 16)
grads = {}
for dt in (torch.float32, torch.float64):
    onet = OracleActorCritic(**CNN_POLICY, dtype=dt)
    onet.load_state_dict(init)
    st, _ = OracleMappo(onet, **ATARI_TRAINER).step(arrays)
    grads[dt] = {k: p.grad.double().numpy().copy() for k, p in onet.params.items()}
    print(dt, {k: st[k] for k in ("policy_loss", "value_loss", "entropy", "grad_norm")})
res = trainer.step(synthetic.to_sample_batch(arrays))
print("hip", {k: res.stats[k] for k in ("policy_loss", "value_loss", "entropy", "grad_norm")})
net = trainer.policy.net
g_hip = net.flat_to_reference(net.grad.detach().cpu())
print(f"{'tensor':52s} {'rms g':>10s} {'hip max':>9s} {'hip rms':>9s} {'o32 max':>9s} {'o32 rms':>9s}   (errors vs float64, / rms g)")
for k, g64 in grads[torch.float64].items():
    rms = np.sqrt((g64**2).mean())
    eh = g_hip[k].numpy().astype(np.float64) - g64
    eo = grads[torch.float32][k] - g64
    print(f"{k:52s} {rms:10.3e} {np.abs(eh).max() / rms:9.2e} {np.sqrt((eh**2).mean()) / rms:9.2e} {np.abs(eo).max() / rms:9.2e} "
          f"{np.sqrt((eo**2).mean()) / rms:9.2e}")
