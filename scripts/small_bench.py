"""BASELINE configs[0] shapes (CartPole: 8 envs x 32 steps, separate 2x64 MLP): one trainer step is ~70 tiny
launches -- this measures how launch-bound it is (ms per step, launches per step, summed kernel time)."""
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
POLICY = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False,
              layernorm=False, shared_backbone=False, seed=1)
for T, B in ((32, 8), (32, 512)):
    tr = trainer_api.make(config.Trainer("mappo", args=dict(popart=False, optimizer_config=dict(lr=3e-4),
                                                           use_graph=bool(int(os.environ.get("SRL_GRAPH", "0"))))),
                          config.Policy("actor-critic", args=POLICY))
    arr = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2)
    if int(os.environ.get("SRL_HOST_SAMPLE", "0")):  # a numpy sample, as the reference's buffer hands it over
        sample = synthetic.to_sample_batch(arr)
    else:
        sample = synthetic.to_sample_batch({k: torch.from_numpy(v).to("cuda:0") for k, v in arr.items()})
    for _ in range(5):
        tr.step(sample)
    torch.cuda.synchronize()
    K = 50
    t0 = time.perf_counter()
    for _ in range(K):
        res = tr.step(sample)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    prof = hip.KernelProfile()
    hip.set_profile(prof)
    tr.step(sample)
    hip.set_profile(None)
    summ = prof.summary()
    print(f"T={T} B={B}: {dt * 1e3:.3f} ms/step ({T * B / dt / 1e3:.1f} k env-steps/s); "
          f"{sum(v['calls'] for v in summ.values())} profiled launches, kernel time {sum(v['ms'] for v in summ.values()):.3f} ms; "
          f"policy_loss {res.stats['policy_loss']:.6f}")
