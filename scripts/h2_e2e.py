# quick end-to-end check of the h2 block inside the trainer: same sample through SRL_H2 on / off
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/srl_amd") else ".")
import srl_amd
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic
from srl_amd.algorithm import h2path
srl_amd.register_all()
POLICY = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
              popart=False, layernorm=False, shared_backbone=True, seed=1,
              cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))
TRAINER = dict(discount_rate=0.99, gae_lambda=0.97, eps_clip=0.2, clip_value=True, dual_clip=False, value_loss='huber',
               value_loss_weight=1.0, value_loss_config=dict(delta=10.0), entropy_bonus_weight=0.01, optimizer='adam',
               optimizer_config=dict(lr=5e-4), popart=False, max_grad_norm=40.0, bootstrap_steps=1)
T, B = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 64
arrays = synthetic.make_sample_arrays(seed=3, T=T, B=B, obs_spec={"obs": ((4, 84, 84), "u8")}, action_dims=6, p_done=0.01)
def run(on):
    h2path.ENABLED = on
    tr = trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, chunk_rows=512)), config.Policy("actor-critic", args=POLICY))
    outs = []
    for _ in range(2):
        r = tr.step(synthetic.to_sample_batch({k: torch.from_numpy(v).to("cuda:0") for k, v in arrays.items()}))
        outs.append(dict(r.stats))
    return outs, tr.policy.net.flat.clone()
from srl_amd import hip
hip.dispatch_counts(reset=True)
a, fa = run(False)
print("off dispatch", hip.dispatch_counts(reset=True))
b, fb = run(True)
print("on  dispatch", hip.dispatch_counts(reset=True))
for i in range(2):
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "clip_ratio"):
        print(i, k, a[i][k], b[i][k], abs(a[i][k] - b[i][k]) / max(abs(a[i][k]), 1e-9))
d = (fa - fb).abs()
print("params max abs diff", float(d.max()), "rms", float((d.double() ** 2).mean().sqrt()), "max |p|", float(fa.abs().max()))
