"""Sequence of profiled launches of one CartPole-shaped trainer step (BASELINE configs[0]: 8 envs x 32 steps, separate 2x64
MLPs), and its time with / without the captured graph."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import srl_amd
from srl_amd import hip
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic
srl_amd.register_all()
POLICY = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False,
              layernorm=False, shared_backbone=False, seed=1)
T, B = 32, 8
for graph in (False, True):
    tr = trainer_api.make(config.Trainer("mappo", args=dict(popart=False, optimizer_config=dict(lr=3e-4), use_graph=graph)),
                          config.Policy("actor-critic", args=POLICY))
    arr = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2)
    sample = synthetic.to_sample_batch({k: torch.from_numpy(v).to("cuda:0") for k, v in arr.items()})
    for _ in range(5):
        tr.step(sample)
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        res = tr.step(sample)
    torch.cuda.synchronize()
    print(f"graph={graph}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step")
    if not graph:
        prof = hip.KernelProfile()
        hip.set_profile(prof)
        tr.step(sample)
        hip.set_profile(None)
        torch.cuda.synchronize()
        print(len(prof.records), "profiled launches:")
        print(" ".join(r[0] for r in prof.records))
