"""BASELINE configs[3] at full size (1024 shared 3m environments x 3 agents x 100 steps, smac_rnn): ms per trainer.step with the
sample on the device.  SRL_RNN_SEQ=0: the per-step recurrent path (one GEMM + one cell kernel per time step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, srl_amd
srl_amd.register_all()
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic
Ts, Bs, A, H = 100, 1024, 3, 64
pol = dict(map_name="3m", hidden_dim=H, chunk_len=10, seed=1, shared=True)
tr_args = dict(popart=True, clip_value=True, dual_clip=False, value_loss="huber", value_loss_config=dict(delta=10.0),
               max_grad_norm=10.0, optimizer_config=dict(lr=5e-4, eps=1e-5))
arrays = synthetic.make_multiagent_arrays(seed=4, T=Ts, B=Bs, agents=A, obs_spec={"local_obs": ((30,), "f32"), "state": ((48,), "f32")},
                                          action_dim=9, p_done=1 / 60, policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
tr = trainer_api.make(config.Trainer("mappo", args=tr_args), config.Policy("smac_rnn", args=pol))
sample = synthetic.to_sample_batch({k: torch.from_numpy(v).to("cuda:0") for k, v in arrays.items()})
for _ in range(3):
    tr.step(sample)
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
a.record()
for _ in range(10):
    tr.step(sample)
e.record()
torch.cuda.synchronize()
print(f"smac_rnn full-size step: {a.elapsed_time(e) / 10:.3f} ms  (SRL_RNN_SEQ={os.environ.get('SRL_RNN_SEQ', '1')})", flush=True)
