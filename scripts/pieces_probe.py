import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import srl_amd
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic
srl_amd.register_all()
name, pargs = "football-smm-separate", dict(obs_dim={"obs": (4, 24, 20)}, hidden_dim=32, seed=3, chunk_len=4, rnn_type="lstm")
arrays = synthetic.make_sample_arrays(seed=5, T=8, B=6, obs_spec={"obs": ((4, 24, 20), "u8")}, action_dims=19, p_done=0.1, policy_state={"actor_hx": (1, 64), "critic_hx": (1, 64)})
targs = dict(popart=True, optimizer_config=dict(lr=5e-4), max_grad_norm=10.0)
for rep in range(3):
    whole = trainer_api.make(config.Trainer("mappo", args=targs), config.Policy(name, args=pargs))
    pieces = trainer_api.make(config.Trainer("mappo", args=targs), config.Policy(name, args=pargs))
    pieces.policy.net.encoder_rows = 13
    r1 = whole.step(synthetic.to_sample_batch({k: v.copy() for k, v in arrays.items()}))
    r2 = pieces.step(synthetic.to_sample_batch({k: v.copy() for k, v in arrays.items()}))
    d = (whole.policy.net.flat - pieces.policy.net.flat).abs()
    gd = (whole.policy.net.grad - pieces.policy.net.grad).abs()
    print(os.environ.get("SRL_WGRAD_STREAM"), "max d", float(d.max()), "frac>1e-6", float((d > 1e-6).float().mean()), "grad max diff", float(gd.max()), "grad rms", float(whole.policy.net.grad.pow(2).mean().sqrt()), r1.stats["grad_norm"], r2.stats["grad_norm"])
