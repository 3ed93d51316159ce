"""BASELINE configs[4] at its per-GPU size (Google Football 11v11: 2048 envs x 200 steps over 8 GPUs = 256 envs per
GPU): the `football-smm-separate` preset with an LSTM (CNN + LSTM policy on stacked (4, 96, 72) uint8 super-mini-map
frames, separate actor / critic backbones, PopArt), synthetic data.  676 M parameters (the preset's default
convolution stack ends in a 22528 -> 11264 Linear per backbone).  Benchmark shortcut: the orthogonal
re-initialisation (a QR of that matrix, minutes on one core) is replaced by a scaled normal draw -- random-init
weights of the same architecture.  FB_B = environments on this GPU, FB_T = steps."""
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _cheap_orthogonal(t, gain=1.0):
    with torch.no_grad():
        return t.normal_(0.0, gain / math.sqrt(t.shape[1] if t.dim() > 1 else t.numel()))


torch.nn.init.orthogonal_ = _cheap_orthogonal
import srl_amd
from srl_amd import hip
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic

srl_amd.register_all()
T, B, H = int(os.environ.get("FB_T", 200)), int(os.environ.get("FB_B", 256)), 128
t0 = time.perf_counter()
tr = trainer_api.make(config.Trainer("mappo", args=dict(popart=True, clip_value=True, value_loss="huber",
                                                        value_loss_config=dict(delta=10.0), max_grad_norm=10.0,
                                                        optimizer_config=dict(lr=5e-4, eps=1e-5))),
                      config.Policy("football-smm-separate", args=dict(rnn_type="lstm", seed=1)))
net = tr.policy.net
print(f"{net.spec.total_params / 1e6:.1f} M parameters, built in {time.perf_counter() - t0:.1f} s; encoder pieces of "
      f"{net.encoder_rows} rows", flush=True)
arr = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec={}, action_dims=19, p_done=1 / 400,
                                   policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
dev = {k: torch.from_numpy(v).to("cuda:0") for k, v in arr.items()}
gen = torch.Generator(device="cuda").manual_seed(0)
dev["obs.obs"] = torch.randint(0, 256, (T + 1, B, 4, 96, 72), dtype=torch.uint8, device="cuda", generator=gen)
sample = synthetic.to_sample_batch(dev)
res = tr.step(sample)
torch.cuda.synchronize()
K = int(os.environ.get("FB_STEPS", 3))
t0 = time.perf_counter()
for _ in range(K):
    res = tr.step(sample)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
flop = 6 * 2 * (92 * 68 * 4 * 100 + 90 * 66 * 8 * 36 + 88 * 64 * 4 * 72 + 22528 * 11264 + 11264 * 5632 + 5632 * 2816 +
                2816 * 1408 + 1408 * 704 + 704 * 128) * T * B
print(f"T={T} B={B}: {dt * 1e3:.1f} ms/step, {T * B / dt / 1e3:.1f} k env-steps/s, ~{flop / dt / 1e12:.1f} TFLOP/s on the "
      f"encoders' contractions, workspace {net.ws.nbytes() / 2**30:.1f} GiB, policy_loss {res.stats['policy_loss']:.5f}, "
      f"grad_norm {res.stats['grad_norm']:.4f}")
prof = hip.KernelProfile()
hip.set_profile(prof)
tr.step(sample)
hip.set_profile(None)
summ = prof.summary()
tot = sum(v["ms"] for v in summ.values())
print(f"kernel time {tot:.1f} ms in {sum(v['calls'] for v in summ.values())} launches")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:10]:
    print(f"  {v['ms']:9.2f} ms  calls={v['calls']:4d}  {k}")
hip.dispatch_tiles(reset=True)
tr.step(sample)
torch.cuda.synchronize()
print("launches per kernel instantiation:", hip.dispatch_tiles(reset=True))
