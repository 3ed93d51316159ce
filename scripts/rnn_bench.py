"""BASELINE configs[3] (SMAC 3m MAPPO): 1024 shared environments x 3 agents, T = 100, the `smac_rnn` policy (separate
LSTM-64 actor / critic over 30- / 48-dim vectors, 9 masked actions, dead-agent masking, PopArt) on
[Tb, B, agents, ...] samples.  Synthetic data; prints agent-steps/s through the GAE+PPO update and the per-kernel
breakdown.  RNN_B = number of environments."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srl_amd
from srl_amd import hip
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic

srl_amd.register_all()
T, B, A, H = 100, int(os.environ.get("RNN_B", 1024)), 3, 64
POLICY = dict(map_name="3m", hidden_dim=H, chunk_len=10, seed=1, shared=True)
TRAINER = dict(popart=True, clip_value=True, value_loss="huber", value_loss_config=dict(delta=10.0), max_grad_norm=10.0,
               optimizer_config=dict(lr=5e-4, eps=1e-5))
tr = trainer_api.make(config.Trainer("mappo", args=TRAINER), config.Policy("smac_rnn", args=POLICY))
arr = synthetic.make_multiagent_arrays(seed=0, T=T, B=B, agents=A,
                                       obs_spec={"local_obs": ((30,), "f32"), "state": ((48,), "f32")}, action_dim=9,
                                       p_done=1 / 60, policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
dev = {k: torch.from_numpy(v).to("cuda:0") for k, v in arr.items()}
sample = synthetic.to_sample_batch(dev)
for _ in range(3):
    tr.step(sample)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    res = tr.step(sample)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"T={T} envs={B} agents={A}: {dt * 1e3:.2f} ms/step, {T * B * A / dt / 1e6:.3f} M agent-steps/s "
      f"({T * B / dt / 1e6:.3f} M env-steps/s), policy_loss {res.stats['policy_loss']:.5f}")
prof = hip.KernelProfile()
hip.set_profile(prof)
tr.step(sample)
hip.set_profile(None)
summ = prof.summary()
tot = sum(v["ms"] for v in summ.values())
print(f"kernel time {tot:.2f} ms in {sum(v['calls'] for v in summ.values())} launches")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:12]:
    print(f"  {v['ms']:8.3f} ms  calls={v['calls']:4d}  {k}")
