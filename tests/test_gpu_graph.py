"""hipGraph capture of the trainer step's device part (use_graph=True): replayed steps equal eager steps."""
import numpy as np
import pytest
import torch

import srl_amd
from srl_amd.api import config, trainer as trainer_api
from srl_amd.runtime import synthetic

srl_amd.register_all()
pytestmark = pytest.mark.gpu

CASES = {
    "mlp": (dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False,
                 layernorm=False, shared_backbone=False, seed=1),
            dict(popart=False, optimizer_config=dict(lr=3e-4), max_grad_norm=0.5),
            dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05)),
    "popart2": (dict(obs_dim=4, action_dim=[3, 2], hidden_dim=32, num_dense_layers=1, num_rnn_layers=0, popart=True,
                     layernorm=True, shared_backbone=True, seed=2),
                dict(popart=True, ppo_epochs=2, clip_value=True, value_loss="huber", value_loss_config=dict(delta=10.0),
                     optimizer_config=dict(lr=1e-3)),
                dict(T=16, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1)),
    "gru": (dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=1, popart=False,
                 layernorm=True, shared_backbone=True, chunk_len=8, seed=3),
            dict(popart=False, optimizer_config=dict(lr=1e-3)),
            dict(T=32, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.08, policy_state={"hx": (1, 32)})),
    # convolution encoder on uint8 frames: the first-layer kernels (byte staging, per-position sums zeroed by a
    # memset node inside the capture) and the grouped data gradient under replay
    "cnn": (dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=64, num_dense_layers=0, num_rnn_layers=0,
                 popart=False, layernorm=False, shared_backbone=True, seed=5,
                 cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')])),
            dict(popart=False, optimizer_config=dict(lr=5e-4), max_grad_norm=40.0, clip_value=True, value_loss="huber",
                 value_loss_config=dict(delta=10.0)),
            dict(T=4, B=3, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.1)),
    # padded and max-pooled convolutions (the general encoder path: explicit LayerNorm, pad / pool / crop kernels)
    "cnnpool": (dict(obs_dim={"img": (4, 22, 18)}, action_dim=4, hidden_dim=32, num_dense_layers=1, num_rnn_layers=0,
                     popart=False, layernorm=True, shared_backbone=True, seed=6, use_maxpool=dict(img=True),
                     cnn_layers=dict(img=[(8, 3, 1, 1, 'zeros'), (8, 3, 1, 1, 'zeros'), (8, 3, 1, 0, 'zeros')])),
                dict(popart=False, optimizer_config=dict(lr=5e-4), max_grad_norm=10.0),
                dict(T=4, B=3, obs_spec={"img": ((4, 22, 18), "u8")}, action_dims=4, p_done=0.1)),
}


@pytest.mark.parametrize("tag", list(CASES))
@pytest.mark.parametrize("resident", [False, True])
def test_graph_replay_equals_eager(tag, resident):
    pargs, targs, skw = CASES[tag]
    mk = lambda graph: trainer_api.make(config.Trainer("mappo", args=dict(targs, use_graph=graph)),
                                        config.Policy("actor-critic", args=pargs))
    eager, graphed = mk(False), mk(True)
    for step in range(5):  # step 0 eager in both, step 1 captures, steps 2.. replay
        arrays = synthetic.make_sample_arrays(seed=50 + step, **skw)
        if resident:
            arrays = {k: torch.from_numpy(v).cuda() for k, v in arrays.items()}
        sa, sb = synthetic.to_sample_batch(dict(arrays)), synthetic.to_sample_batch(dict(arrays))
        ra, rb = eager.step(sa), graphed.step(sb)
        assert ra.step == rb.step
        for k, v in ra.stats.items():
            # two trainers: float64 atomics order the last bits of the statistics, float32 sums follow
            assert abs(v - rb.stats[k]) <= 2e-5 * max(1.0, abs(v)), (tag, step, k, v, rb.stats[k])
        adv_a, adv_b = sa.analyzed_result.adv, sb.analyzed_result.adv
        to_np = lambda x: x.cpu().numpy() if isinstance(x, torch.Tensor) else x
        assert np.allclose(to_np(adv_a), to_np(adv_b), rtol=1e-6, atol=1e-7)
    pa, pb = eager.get_checkpoint(), graphed.get_checkpoint()
    for k in pa["state_dict"]:
        assert torch.allclose(pa["state_dict"][k], pb["state_dict"][k], rtol=0, atol=1e-5), k
    st_a, st_b = pa["optimizer_state_dict"]["state"], pb["optimizer_state_dict"]["state"]
    assert float(st_a[0]["step"]) == float(st_b[0]["step"])
    assert len(graphed._graphs) == 1 and next(iter(graphed._graphs.values())) is not None
