"""BASELINE.json configs[1] at FULL size (512 envs x 128 steps, uint8 (4,84,84) frames, NatureCNN-512): the oracle
cannot run this in seconds, so parity is checked through size-independent properties of the path."""
import numpy as np
import pytest
import torch

import srl_amd
from srl_amd import hip
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic

srl_amd.register_all()
pytestmark = pytest.mark.gpu

POLICY = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
              popart=False, layernorm=False, shared_backbone=True, seed=1,
              cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))
TRAINER = dict(discount_rate=0.99, gae_lambda=0.97, eps_clip=0.2, clip_value=True, dual_clip=False, value_loss='huber',
               value_loss_weight=1.0, value_loss_config=dict(delta=10.0), entropy_bonus_weight=0.01,
               optimizer_config=dict(lr=5e-4), popart=False, max_grad_norm=40.0)
T, B = 128, 512


def device_sample(seed):
    arr = synthetic.make_sample_arrays(seed=seed, T=T, B=B, obs_spec={}, action_dims=6, p_done=1.0 / 800)
    dev = {k: torch.from_numpy(v).cuda() for k, v in arr.items()}
    gen = torch.Generator(device="cuda").manual_seed(seed)
    dev["obs.obs"] = torch.randint(0, 256, (T + 1, B, 4, 84, 84), dtype=torch.uint8, device="cuda", generator=gen)
    return dev


def make(chunk_rows):
    return trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, chunk_rows=chunk_rows)),
                            config.Policy("actor-critic", args=POLICY))


def test_gae_scan_full_size_properties():
    """Linearity (adv is linear in (reward, value) for fixed flags), agreement of the returned masked sums with a
    float64 recomputation from the returned advantages, and ret = adv + value*(1-done) -- at 128 x 512 and 128 x 4096."""
    for b in (512, 4096):
        arr = synthetic.make_sample_arrays(seed=b, T=T, B=b, obs_spec={}, action_dims=2, p_done=1.0 / 800)
        d = {k: torch.from_numpy(v).cuda() for k, v in arr.items()}

        def scan(r, v):
            adv, ret = torch.zeros((T + 1, b, 1), device="cuda"), torch.zeros((T + 1, b, 1), device="cuda")
            stats = torch.zeros(3, dtype=torch.float64, device="cuda")
            hip.gae_scan(r, v, d["done"], d["truncated"], d["on_reset"], 0.99, 0.97, adv, ret, stats=stats)
            return adv, ret, stats

        r, v = d["reward"], d["analyzed_result.value"]
        a1, ret1, s1 = scan(r, v)
        a2, _, _ = scan(2.5 * r, 2.5 * v)
        assert torch.allclose(a2, 2.5 * a1, rtol=1e-5, atol=1e-5)
        mask = (1 - d["on_reset"][1:].double())
        x = a1[:T].double() * mask
        assert s1[0].item() == mask.sum().item()
        assert abs(s1[1].item() - x.sum().item()) <= 1e-9 * max(1.0, x.abs().sum().item())
        assert abs(s1[2].item() - (x * x).sum().item()) <= 1e-9 * (x * x).sum().item()
        vm = v * (1 - d["done"].float())
        assert torch.equal(ret1[:T], a1[:T] + vm[:T])  # float32 add, as mappo.py:143


def test_step_full_size_chunking_and_repeatability():
    """The row-chunking of the forward/backward (activation workspace) must not change the update; the same step on
    the same weights must repeat; returned advantages obey the masked statistics reported in the stats."""
    sample = device_sample(7)
    results = []
    for chunk in (16384, 8192, 16384):
        tr = make(chunk)
        res = tr.step(synthetic.to_sample_batch(dict(sample)))
        sd = tr.policy.get_checkpoint()["state_dict"]
        results.append((res.stats, sd))
    (s0, p0), (s1, p1), (s2, p2) = results
    for k in s0:
        tol = 1e-5 if k in ("policy_loss", "value_loss", "entropy") else 1e-4
        assert abs(s0[k] - s1[k]) <= tol * max(abs(s0[k]), 1e-3), ("chunking", k, s0[k], s1[k])
        assert abs(s0[k] - s2[k]) <= 1e-6 * max(abs(s0[k]), 1e-3), ("repeat", k, s0[k], s2[k])
    for k in p0:
        assert torch.allclose(p0[k], p1[k], rtol=0, atol=2e-5), ("chunking", k)
        assert torch.allclose(p0[k], p2[k], rtol=0, atol=1e-6), ("repeat", k)
    assert s0["frames"] == T * B and np.isfinite(list(s0.values())).all()
