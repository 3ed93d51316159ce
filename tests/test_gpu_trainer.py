"""GPU parity tests of the whole drop-in path: policy.analyze / rollout and trainer.step through the plugin API,
against the golden vectors generated from the real reference and against the CPU oracle at larger sizes."""
import copy

import numpy as np
import pytest
import torch

import srl_amd
from oracle.net import OracleActorCritic
from oracle.trainer import OracleMappo
from srl_amd.api import config, policy as policy_api, trainer as trainer_api
from srl_amd.namedarray import NamedArray
from srl_amd.runtime import synthetic

srl_amd.register_all()
pytestmark = pytest.mark.gpu

C1_POLICY = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False,
                 layernorm=False, shared_backbone=False, chunk_len=8, seed=1)
ATARI_TRAINER = dict(discount_rate=0.99, gae_lambda=0.97, eps_clip=0.2, clip_value=True, dual_clip=False,
                     value_loss='huber', value_loss_weight=1.0, value_loss_config=dict(delta=10.0),
                     entropy_bonus_weight=0.01, optimizer='adam', optimizer_config=dict(lr=5e-4), popart=False,
                     max_grad_norm=40.0, bootstrap_steps=1)
CNN_POLICY = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
                  popart=False, layernorm=False, shared_backbone=True, chunk_len=4, seed=5,
                  cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))
CASES = {
    "c1": (C1_POLICY, dict(popart=False, optimizer_config=dict(lr=3e-4)),
           dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05), 3, "steps_mlp.npz"),
    "c1atari": (C1_POLICY, ATARI_TRAINER, dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05), 2,
                "steps_mlp.npz"),
    "c1ln": (dict(C1_POLICY, layernorm=True, seed=2),
             dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=0.5),
             dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05), 2, "steps_mlp.npz"),
    "multi": (dict(obs_dim={"a": 5, "b": 3}, action_dim=[3, 4], hidden_dim=32, num_dense_layers=1, num_rnn_layers=0,
                   popart=False, layernorm=True, shared_backbone=True, chunk_len=8, seed=3, activation="tanh"),
              dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=1e-3)),
              dict(T=16, B=4, obs_spec={"a": ((5,), "f32"), "b": ((3,), "f32")}, action_dims=[3, 4], p_done=0.1), 2,
              "steps_mlp.npz"),
    "cnn": (CNN_POLICY, ATARI_TRAINER, dict(T=4, B=3, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.1), 2,
            "steps_cnn.npz"),
    # PopArt value head (the reference policy's default) and V-trace through the trainer (gen_golden.py gen_popart)
    "pa": (dict(C1_POLICY, popart=True, seed=7), dict(popart=True, optimizer_config=dict(lr=3e-4)),
           dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05), 3, "steps_popart.npz"),
    "pa2": (dict(C1_POLICY, popart=True, layernorm=True, shared_backbone=True, seed=8),
            dict(popart=True, clip_value=True, dual_clip=False, value_loss='huber', value_loss_config=dict(delta=10.0),
                 value_loss_weight=1.0, ppo_epochs=2, optimizer_config=dict(lr=5e-4), max_grad_norm=40.0),
            dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05), 2, "steps_popart.npz"),
    "vt": (dict(C1_POLICY, seed=9), dict(popart=False, vtrace=True, optimizer_config=dict(lr=3e-4)),
           dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05, p_trunc=0.0), 2,
           "steps_popart.npz"),
    "vtpa": (dict(C1_POLICY, popart=True, seed=10),
             dict(popart=True, vtrace=True, max_grad_norm=10.0, optimizer_config=dict(lr=1e-3)),
             dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05, p_trunc=0.0), 2,
             "steps_popart.npz"),
    # recurrent backbones: GRU + auto reset, chunked analysis from the stored states (gen_golden.py gen_rnn)
    "gru": (dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=1, popart=False,
                 layernorm=True, shared_backbone=True, chunk_len=8, seed=21),
            dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0),
            dict(T=32, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.08, policy_state={"hx": (1, 32)}),
            3, "steps_rnn.npz"),
    "gru2": (dict(obs_dim=4, action_dim=[3, 2], hidden_dim=16, num_dense_layers=2, num_rnn_layers=2, popart=True,
                  layernorm=False, shared_backbone=False, chunk_len=16, seed=22),
             dict(popart=True, ppo_epochs=2, optimizer_config=dict(lr=5e-4)),
             dict(T=16, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1,
                  policy_state={"actor_hx": (2, 16), "critic_hx": (2, 16)}), 2, "steps_rnn.npz"),
    "lstm": (dict(obs_dim=4, action_dim=2, hidden_dim=16, num_dense_layers=1, num_rnn_layers=2, rnn_type="lstm",
                  popart=False, layernorm=True, shared_backbone=True, chunk_len=4, seed=23),
             dict(popart=False, optimizer_config=dict(lr=1e-3)),
             dict(T=16, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.1, policy_state={"hx": (2, 32)}),
             2, "steps_rnn.npz"),
    # burn-in: 2 rows before every chunk replayed without gradient (18 stored rows = 2 + 4 chunks of 4)
    "burn": (dict(obs_dim=4, action_dim=2, hidden_dim=16, num_dense_layers=1, num_rnn_layers=1, popart=False,
                  layernorm=True, shared_backbone=False, chunk_len=4, seed=24),
             dict(popart=False, burn_in_steps=2, optimizer_config=dict(lr=1e-3)),
             dict(T=18, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.1,
                  policy_state={"actor_hx": (1, 16), "critic_hx": (1, 16)}), 2, "steps_rnn.npz"),
}
# the other optimisers modules/utils.py:268-286 accepts (gen_golden.py gen_optim)
_C1S = dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05)
CASES.update({
    "rms": (dict(C1_POLICY, seed=41), dict(popart=False, optimizer='rmsprop', optimizer_config=dict(lr=1e-3)), _C1S, 3,
            "steps_optim.npz"),
    "rmsc": (dict(C1_POLICY, layernorm=True, seed=42),
             dict(popart=False, optimizer='rmsprop', max_grad_norm=5.0,
                  optimizer_config=dict(lr=5e-4, alpha=0.95, eps=1e-6, momentum=0.9, centered=True, weight_decay=1e-3)), _C1S, 3,
             "steps_optim.npz"),
    "sgd": (dict(C1_POLICY, seed=43), dict(popart=False, optimizer='sgd', optimizer_config=dict(lr=1e-2)), _C1S, 2,
            "steps_optim.npz"),
    "sgdn": (dict(C1_POLICY, shared_backbone=True, seed=44),
             dict(popart=False, optimizer='sgd', max_grad_norm=1.0,
                  optimizer_config=dict(lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-3)), _C1S, 3, "steps_optim.npz"),
    "sgdd": (dict(C1_POLICY, seed=45),
             dict(popart=False, optimizer='sgd', ppo_epochs=2, optimizer_config=dict(lr=1e-2, momentum=0.8, dampening=0.3)),
             _C1S, 2, "steps_optim.npz"),
})
# V-trace with recurrent policies (gen_golden.py gen_vtrace_rnn)
CASES.update({
    "vtgru": (dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=1, popart=False, layernorm=True,
                   shared_backbone=True, chunk_len=8, seed=51),
              dict(popart=False, vtrace=True, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0),
              dict(T=32, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.08, p_trunc=0.0,
                   policy_state={"hx": (1, 32)}), 2, "steps_vtrace_rnn.npz"),
    "vtlstm": (dict(obs_dim=4, action_dim=[3, 2], hidden_dim=16, num_dense_layers=1, num_rnn_layers=1, rnn_type="lstm",
                    popart=True, layernorm=False, shared_backbone=False, chunk_len=4, seed=52),
               dict(popart=True, vtrace=True, ppo_epochs=2, optimizer_config=dict(lr=5e-4)),
               dict(T=16, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1, p_trunc=0.0,
                    policy_state={"actor_hx": (1, 32), "critic_hx": (1, 32)}), 2, "steps_vtrace_rnn.npz"),
})
CASES["vtb2"] = (dict(C1_POLICY, chunk_len=1, seed=53), dict(popart=False, vtrace=True, bootstrap_steps=2, ppo_epochs=2,
                                                optimizer_config=dict(lr=1e-3)),
                 dict(T=31, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05, p_trunc=0.0, bootstrap_steps=2), 2,
                 "steps_vtrace_rnn.npz")
# value_dim > 1 through the loss (gen_golden.py gen_value_dim: the reference under its launcher's `python -O` semantics)
CASES.update({
    "vd3": (dict(C1_POLICY, value_dim=3, seed=61), dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=5.0),
            dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05, value_dim=3), 2, "steps_value_dim.npz"),
    "vd2pa": (dict(C1_POLICY, action_dim=[3, 2], value_dim=2, popart=True, layernorm=True, shared_backbone=True, seed=62),
              dict(popart=True, ppo_epochs=2, clip_value=True, dual_clip=False, value_loss='huber',
                   value_loss_config=dict(delta=10.0), optimizer_config=dict(lr=5e-4)),
              dict(T=16, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1, value_dim=2), 2,
              "steps_value_dim.npz"),
})
# continuous actions: Normal(mean, std) with the three parametrisations of log sigma (gen_golden.py gen_continuous)
_CBASE = dict(obs_dim=7, action_dim=3, hidden_dim=32, num_dense_layers=2, num_rnn_layers=0, popart=False, layernorm=True,
              shared_backbone=False, chunk_len=8, continuous_action=True)
_CSMP = dict(T=16, B=6, obs_spec={"obs": ((7,), "f32")}, action_dims=3, p_done=0.1, continuous_action=True)
_CTR = dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=5.0)
CASES.update({
    "cfix": (dict(_CBASE, std_type="fixed", init_log_std=-0.3, seed=31), _CTR, _CSMP, 2, "steps_continuous.npz"),
    "csep": (dict(_CBASE, std_type="separate_learnable", seed=32), dict(_CTR, ppo_epochs=2), _CSMP, 2,
             "steps_continuous.npz"),
    "cshr": (dict(_CBASE, std_type="shared_learnable", shared_backbone=True, seed=33), _CTR, _CSMP, 2,
             "steps_continuous.npz"),
})
# zero padding and max-pooling between the convolutions (gen_golden.py gen_cnn_padpool)
PAD_POLICY = dict(obs_dim={"obs": (4, 20, 20)}, action_dim=5, hidden_dim=32, num_dense_layers=1, num_rnn_layers=0,
                  popart=False, layernorm=False, shared_backbone=True, chunk_len=4, seed=71,
                  cnn_layers=dict(obs=[(8, 3, 1, 1, 'zeros'), (16, 3, 2, 1, 'zeros'), (8, 3, 1, 0, 'zeros')]))
POOL_POLICY = dict(obs_dim={"img": (3, 23, 19), "vec": 5}, action_dim=[3, 2], hidden_dim=16, num_dense_layers=1,
                   num_rnn_layers=0, popart=True, layernorm=True, shared_backbone=False, chunk_len=4, seed=72,
                   activation="tanh", use_maxpool=dict(img=True),
                   cnn_layers=dict(img=[(4, 3, 1, 0, 'zeros'), (8, 3, 1, 1, 'zeros'), (4, 3, 1, 0, 'zeros')]))
CASES.update({
    "cnnpad": (PAD_POLICY, dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=5e-4), max_grad_norm=10.0),
               dict(T=6, B=4, obs_spec={"obs": ((4, 20, 20), "u8")}, action_dims=5, p_done=0.1), 2, "steps_cnn_padpool.npz"),
    "cnnpool": (POOL_POLICY, dict(popart=True, optimizer_config=dict(lr=1e-3)),
                dict(T=5, B=3, obs_spec={"img": ((3, 23, 19), "f32"), "vec": ((5,), "f32")}, action_dims=[3, 2], p_done=0.1),
                2, "steps_cnn_padpool.npz"),
})
# observations with one / three spatial dimensions: nn.Conv1d / nn.Conv3d encoders (gen_golden.py gen_cnn_nd)
ND1_POLICY = dict(obs_dim={"seq": (3, 59)}, action_dim=4, hidden_dim=16, num_dense_layers=1, num_rnn_layers=0, popart=False,
                  layernorm=True, shared_backbone=True, chunk_len=4, seed=73, use_maxpool=dict(seq=True),
                  cnn_layers=dict(seq=[(4, 3, 1, 0, 'zeros'), (8, 3, 2, 1, 'zeros'), (4, 3, 1, 0, 'zeros')]))
ND3_POLICY = dict(obs_dim={"vol": (2, 9, 8, 7), "vec": 3}, action_dim=[2, 3], hidden_dim=16, num_dense_layers=1,
                  num_rnn_layers=0, popart=False, layernorm=False, shared_backbone=False, chunk_len=4, seed=74,
                  activation="tanh", use_maxpool=dict(vol=True),
                  cnn_layers=dict(vol=[(4, 3, 1, 1, 'zeros'), (4, 2, 1, 0, 'zeros')]))
CASES.update({
    "cnn1d": (ND1_POLICY, dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0),
              dict(T=6, B=4, obs_spec={"seq": ((3, 59), "f32")}, action_dims=4, p_done=0.1), 2, "steps_cnn_nd.npz"),
    "cnn3d": (ND3_POLICY, dict(popart=False, optimizer_config=dict(lr=1e-3)),
              dict(T=5, B=3, obs_spec={"vol": ((2, 9, 8, 7), "u8"), "vec": ((3,), "f32")}, action_dims=[2, 3], p_done=0.1), 2,
              "steps_cnn_nd.npz"),
})
# nn.ConvNd's non-zero padding modes (gen_golden.py gen_cnn_padmode)
CASES["padm"] = (dict(obs_dim={"img": (4, 12, 10), "seq": (2, 21)}, action_dim=3, hidden_dim=16, num_dense_layers=1,
                      num_rnn_layers=0, popart=False, layernorm=False, shared_backbone=True, chunk_len=4, seed=75,
                      cnn_layers=dict(img=[(8, 3, 1, 2, 'reflect'), (8, 3, 2, 1, 'circular'), (4, 3, 1, 1, 'zeros')],
                                      seq=[(4, 5, 2, 3, 'replicate'), (4, 3, 1, 1, 'circular')])),
                 dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0),
                 dict(T=5, B=4, obs_spec={"img": ((4, 12, 10), "u8"), "seq": ((2, 21), "f32")}, action_dims=3, p_done=0.1), 2,
                 "steps_cnn_padmode.npz")


def make_trainer(policy_args, trainer_args):
    return trainer_api.make(config.Trainer("mappo", args=trainer_args), config.Policy("actor-critic", args=policy_args))


def close(a, b, rtol, scale=1.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool((np.abs(a - b) <= rtol * np.maximum(np.abs(b), scale)).all())


@pytest.mark.parametrize("tag", list(CASES))
def test_step_matches_reference_golden(tag, golden):
    pargs, targs, skw, n_steps, fname = CASES[tag]
    g = golden(fname)
    trainer = make_trainer(pargs, targs)
    names = list(g[f"{tag}_stat_names"])
    for step in range(n_steps):
        arrays = synthetic.make_sample_arrays(seed=100 + step, **skw)
        sample = synthetic.to_sample_batch(arrays)
        if step == 0 and f"{tag}_analyze_new_lp" in g.files:
            Tb = arrays["on_reset"].shape[0]
            ar = trainer.policy.analyze(sample[:Tb - 1], target="ppo")
            assert close(ar.new_action_log_probs.cpu().numpy(), g[f"{tag}_analyze_new_lp"], 1e-5), "analyze log-probs"
            assert close(ar.state_values.cpu().numpy(), g[f"{tag}_analyze_value"], 1e-5), "analyze values"
            assert close(ar.entropy.cpu().numpy(), g[f"{tag}_analyze_entropy"], 1e-5), "analyze entropy"
        res = trainer.step(sample)
        ref = dict(zip(names, g[f"{tag}_step{step}_stats"]))
        for k in ("policy_loss", "value_loss", "entropy", "advantage", "value_targets", "importance_weight", "clip_ratio",
                  "done", "truncated", "grad_norm", "frames"):
            tol = 1e-5 if k in ("policy_loss", "value_loss", "entropy", "value_targets") else 1e-4
            assert abs(res.stats[k] - ref[k]) <= tol * max(abs(ref[k]), 1e-2), (tag, step, k, res.stats[k], ref[k])
        if "denorm_value" in ref:  # PopArt: masked mean of the de-normalised value targets
            assert abs(res.stats["denorm_value"] - ref["denorm_value"]) <= 1e-5 * max(abs(ref["denorm_value"]), 1e-2)
        if step == 0:  # GAE returns written back into the sample: 1e-5 relative (BASELINE.json)
            assert close(sample.analyzed_result.adv, g[f"{tag}_step0_adv"], 1e-5)
            assert close(sample.analyzed_result.ret, g[f"{tag}_step0_ret"], 1e-5)
        if step in (0, n_steps - 1):
            sd = trainer.policy.get_checkpoint()["state_dict"]
            for key in g.files:
                pre_full, pre_s = f"{tag}_step{step}_param:", f"{tag}_step{step}_param_s97:"
                if key.startswith(pre_full):
                    got = sd[key[len(pre_full):]].numpy()
                elif key.startswith(pre_s):
                    got = sd[key[len(pre_s):]].numpy().reshape(-1)[::97]
                else:
                    continue
                # parameters after Adam steps: each step moves a weight by ~lr, so compare at a few % of lr
                assert np.abs(got - g[key]).max() <= 2e-5, (tag, step, key, np.abs(got - g[key]).max())
                if "_RunningMeanStd__" in key:  # float64 PopArt statistics: EMA of float64 masked sums
                    assert got.dtype == np.float64 and np.allclose(got, g[key], rtol=1e-6, atol=1e-13), (tag, step, key)
    assert trainer.policy.version == int(g[f"{tag}_version"])
    assert res.step == trainer.policy.version


def test_init_matches_reference_init(golden):
    """Same seed -> the reference's initial weights (bit-identical in the container that made the fixtures, where
    gen_golden.py asserts torch.equal; LAPACK's QR may differ in the last bit on another CPU, hence allclose)."""
    g = golden("steps_mlp.npz")
    for tag in ("c1", "c1ln", "multi"):
        pol = policy_api.make(config.Policy("actor-critic", args=CASES[tag][0]))
        for k, v in pol.get_checkpoint()["state_dict"].items():
            assert np.allclose(v.numpy(), g[f"{tag}_init_param:{k}"], rtol=1e-4, atol=1e-4), (tag, k)


def test_rollout_eval_golden(golden):
    g = golden("rollout.npz")
    pol = policy_api.make(config.Policy("actor-critic", args=C1_POLICY))
    N = g["c1_obs"].shape[0]
    req = policy_api.RolloutRequest(obs=NamedArray(obs=g["c1_obs"]), is_evaluation=np.ones((N, 1), np.uint8),
                                    on_reset=np.zeros((N, 1), np.uint8))
    res = pol.rollout(req)
    assert res.action.x.dtype == np.int64 and res.action.x.shape == (N, 1)
    assert np.array_equal(res.action.x, g["c1_action"])
    assert close(res.analyzed_result.log_probs, g["c1_log_probs"], 1e-5)
    assert close(res.analyzed_result.value, g["c1_value"], 1e-5, scale=1e-2)
    assert res.policy_state is None


@pytest.mark.parametrize("T,B,chunk", [(128, 64, 16384), (64, 33, 1000)])
def test_step_vs_oracle_larger(T, B, chunk):
    """Bigger than the fixtures, incl. several row-chunks with a ragged tail: loss terms and GAE vs the CPU oracle."""
    pargs = dict(C1_POLICY, layernorm=True, seed=7)
    targs = dict(ATARI_TRAINER, chunk_rows=chunk)
    trainer = make_trainer(pargs, targs)
    onet = OracleActorCritic(**pargs)
    onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **{k: v for k, v in targs.items() if k != "chunk_rows"})
    for step in range(2):
        arrays = synthetic.make_sample_arrays(seed=step, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2,
                                              p_done=0.02)
        sample = synthetic.to_sample_batch(arrays)
        res = trainer.step(sample)
        ostats, oout = oracle.step(arrays)
        assert close(sample.analyzed_result.ret, oout["ret"], 1e-5)
        for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
            assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (step, k, res.stats[k], ostats[k])
    sd = trainer.policy.get_checkpoint()["state_dict"]
    osd = onet.state_dict()
    for k in sd:
        assert np.abs(sd[k].numpy() - osd[k].numpy()).max() <= 3e-5, k


@pytest.mark.parametrize("kind", ["vector", "image"])
def test_wide_dense_layers_on_presplit_operands_vs_oracle(kind, monkeypatch):
    """`HipNet._linear_fwd_h2d / _linear_bwd_h2d` (round 6): a Linear with >= 1024 inputs over >= 4096 rows runs its three
    products on the pre-split kernels (srl_h2_pack_rows + srl_h2_gemm / srl_h2_wgrad_dense) -- the football preset's dense tower.
    `vector`: LayerNorm -> Linear 1056 -> 256 (no ReLU mask below it); `image`: the reference's default convolution stack
    (cnn.py:96-98: 4 / 8 / 4 channels -- no sign words from the convolutions, `srl_relu_mask` makes them) in front of
    Linear 1024 -> 512 + the rest of the halving tower.  One step against the CPU oracle (returns, loss terms, gradient norm, every
    tensor's gradient to 2 % of its rms), and both this step's and the layer-by-layer kernels' (SRL_H2_DENSE=0) gradients against the
    float64 oracle: every tensor's error at most 3 x the layer-by-layer kernels' (+ 2e-6 of the tensor's largest element)."""
    from srl_amd import hip
    from srl_amd.algorithm.hipnet import HipNet
    if kind == "vector":
        pargs = dict(obs_dim=1056, action_dim=5, hidden_dim=256, num_dense_layers=1, num_rnn_layers=0, popart=False, layernorm=True,
                     shared_backbone=False, chunk_len=8, seed=31)
        spec, T, B = {"obs": ((1056,), "f32")}, 64, 80
    else:
        pargs = dict(obs_dim={"obs": (4, 24, 24)}, action_dim=5, hidden_dim=64, num_dense_layers=0, num_rnn_layers=0, popart=False,
                     layernorm=True, shared_backbone=True, chunk_len=8, seed=32,
                     cnn_layers=dict(obs=[(4, 5, 1, 0, 'zeros'), (8, 3, 1, 0, 'zeros'), (4, 3, 1, 0, 'zeros')]))
        spec, T, B = {"obs": ((4, 24, 24), "u8")}, 64, 72
    targs = dict(ATARI_TRAINER, chunk_rows=1 << 20)
    arrays = synthetic.make_sample_arrays(seed=3, T=T, B=B, obs_spec=spec, action_dims=5, p_done=0.02)
    grads, stats = {}, {}
    for dense in (True, False):
        monkeypatch.setattr(HipNet, "H2_DENSE", dense)
        trainer = make_trainer(pargs, targs)
        net = trainer.policy.net
        if dense:
            onet = OracleActorCritic(**pargs)
            onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
            oracle = OracleMappo(onet, **{k: v for k, v in targs.items() if k != "chunk_rows"})
        sample = synthetic.to_sample_batch(arrays)
        hip.dispatch_tiles(reset=True)
        res = trainer.step(sample)
        tiles = hip.dispatch_tiles(reset=True)
        assert (any(k.startswith("h2:tn") for k in tiles) and any(k.startswith("h2:gemm") for k in tiles)) == dense, tiles
        grads[dense] = {k: v.clone() for k, v in net.flat_to_reference(net.grad.detach().cpu()).items()}
        stats[dense] = res.stats
        if dense:
            ostats, oout = oracle.step(arrays)
            assert close(sample.analyzed_result.ret, oout["ret"], 1e-5)
            for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
                assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (k, res.stats[k], ostats[k])
            for k, p in onet.params.items():
                g_ref = p.grad.double().numpy()
                rms = np.sqrt((g_ref**2).mean())
                err = np.sqrt(((grads[True][k].double().numpy() - g_ref)**2).mean())
                assert err <= 2e-2 * rms, (k, float(err / max(rms, 1e-30)))
    # the two kernel families against the float64 restatement of the same step: the pre-split kernels' gradients are as close to it
    # as the layer-by-layer kernels' (a LayerNorm weight's gradient is a sum over rows that cancels to 1e-2 of its terms: the two
    # float32 paths differ there by 2e-4 of the tensor's largest element, both 1e-4 from float64)
    onet64 = OracleActorCritic(**pargs, dtype=torch.float64)
    onet64.load_state_dict({k: v.numpy() for k, v in make_trainer(pargs, targs).policy.get_checkpoint()["state_dict"].items()})
    OracleMappo(onet64, **{k: v for k, v in targs.items() if k != "chunk_rows"}).step(arrays)
    for k, p in onet64.params.items():
        g64 = p.grad.double()
        e1 = float((grads[True][k].double() - g64).abs().max())
        e0 = float((grads[False][k].double() - g64).abs().max())
        scale = float(g64.abs().max())
        assert e1 <= 3.0 * e0 + 2e-6 * scale, (k, e1, e0, scale)


def test_cnn_step_vs_oracle_chunked():
    """NatureCNN with more samples than one row-chunk, uint8 frames resident on the device."""
    targs = dict(ATARI_TRAINER, chunk_rows=40)
    trainer = make_trainer(CNN_POLICY, targs)
    onet = OracleActorCritic(**CNN_POLICY)
    onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **ATARI_TRAINER)
    arrays = synthetic.make_sample_arrays(seed=11, T=12, B=8, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.1)
    sample = synthetic.to_sample_batch(arrays)
    res = trainer.step(sample)
    ostats, _ = oracle.step(arrays)
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (k, res.stats[k], ostats[k])


def test_separate_actor_and_critic_cnn_on_the_presplit_path_vs_oracle():
    """`shared_backbone=False` (the reference's default `actor-critic` registration, actor_critic_policy.py:146-166) with the
    Atari stack: the actor's and the critic's encoder both read the key "obs" and have weights of their own.  With 288 rows per
    chunk both go through the pre-split block (h2path.py, >= H2_MIN_ROWS rows), each with ITS weights, scales and gradients --
    two chunks per update, so that the second chunk's forward runs over the first one's buffers.  Per step (every step starts
    from the device's parameters on both sides, as in test_cnn_step_at_the_benchmarked_dispatch): GAE returns and loss terms at
    1e-5, the gradient norm at 5e-5, and the gradient of EVERY parameter tensor -- the critic encoder's own included -- against
    the float32 oracle's to 2 % of the tensor's rms (a ReLU unit within rounding of zero flips its mask on either side and
    moves the sums upstream by a row's share; the defect this guards against -- the critic running through the actor's block --
    leaves the critic encoder's gradients ZERO and doubles the actor's)."""
    from srl_amd import hip
    pargs = dict(CNN_POLICY, shared_backbone=False, seed=6)
    trainer = make_trainer(pargs, dict(ATARI_TRAINER, chunk_rows=288))
    net = trainer.policy.net
    onet = OracleActorCritic(**pargs)
    onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **ATARI_TRAINER)
    hip.dispatch_counts(reset=True)
    for step in range(2):
        arrays = synthetic.make_sample_arrays(seed=31 + step, T=18, B=32, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.05)
        before = trainer.policy.get_checkpoint()["state_dict"]
        for k, p in onet.params.items():
            p.data.copy_(before[k].to(p.dtype))
        sample = synthetic.to_sample_batch(arrays)
        res = trainer.step(sample)
        ostats, oout = oracle.step(arrays)
        assert close(sample.analyzed_result.ret, oout["ret"], 1e-5), step
        for k in ("policy_loss", "value_loss", "entropy"):
            assert abs(res.stats[k] - ostats[k]) <= 1e-5 * max(abs(ostats[k]), 1e-2), (step, k, res.stats[k], ostats[k])
        assert abs(res.stats["grad_norm"] - ostats["grad_norm"]) <= 5e-5 * max(abs(ostats["grad_norm"]), 1e-2), step
        g_hip = net.flat_to_reference(net.grad.detach().cpu())
        critic_own = 0
        for k, p in onet.params.items():
            g_ref = p.grad.double().numpy()
            rms = np.sqrt((g_ref**2).mean())
            assert rms > 0, k
            err = np.sqrt(((g_hip[k].double().numpy() - g_ref)**2).mean())
            assert err <= 2e-2 * rms, (step, k, float(err / rms))
            critic_own += int(k.startswith("state_modules_dict."))
        assert critic_own == 12, critic_own  # LayerNorm, three convolutions and two Linears of the critic's own encoder
    counts = hip.dispatch_counts(reset=True)
    assert counts["h2"] == 2 * 2 * 2 * 9, counts  # steps x chunks x encoders x the block's nine launches (round 6: the Linear's weight gradient too)
    assert len([b for b in net._h2_blocks.values() if b is not None]) == 2  # one block per encoder, not per observation key


@pytest.mark.parametrize("tail,pipelines", [(512, 1), (100, 1), (512, 2)])
def test_first_layer_accumulation_with_a_ragged_last_chunk(tail, pipelines, monkeypatch):
    """The chunks of an update share one finalisation of the first layer's gradients (srl_conv2d_obs_bwd `phase`): a last chunk
    shorter than the others -- 512 rows: still on the pre-split block; 100 rows: below H2_MIN_ROWS, on the layer-by-layer path --
    must add to and close what the chunks before it opened.  Against the same step with SRL_OBS_BWD_DEFER=0 (every chunk
    finalises by itself): conv1 / observation LayerNorm gradients to float32 summation-order accuracy."""
    T, B, chunk = 1, 2 * 4096 + tail, 4096
    arrays = synthetic.make_sample_arrays(seed=5, T=T, B=B, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.05)
    grads = {}
    for defer in ("1", "0"):
        monkeypatch.setenv("SRL_OBS_BWD_DEFER", defer)
        trainer = make_trainer(CNN_POLICY, dict(ATARI_TRAINER, chunk_rows=chunk, pipelines=pipelines))
        trainer.step(synthetic.to_sample_batch(arrays))
        net = trainer.policy.net
        grads[defer] = {k: v.clone() for k, v in net.flat_to_reference(net.grad.detach().cpu()).items()}
        assert not any(b.open for x in [net] + list(trainer._twin or []) for b in x._h2_blocks.values() if b is not None)
    for k, g0 in grads["0"].items():
        g1 = grads["1"][k]
        scale = float(g0.abs().max())
        assert float((g1 - g0).abs().max()) <= 2e-5 * max(scale, 1e-8), (k, float((g1 - g0).abs().max()), scale)


def _pong_frames(rng, Tb, B):
    """Atari-`pong`-type frames: a flat bright background (large mean, tiny variance) with a few small sprites -- the
    case in which an un-centred byte contraction loses digits (first convolution on bytes)."""
    f = np.full((Tb, B, 4, 84, 84), 144, dtype=np.uint8)
    for t in range(Tb):
        for b in range(B):
            for c in range(4):
                y, x = rng.integers(0, 76, size=2)
                f[t, b, c, y:y + 8, x:x + 2] = 236
                y, x = rng.integers(0, 80, size=2)
                f[t, b, c, y:y + 2, x:x + 2] = rng.integers(0, 256)
    return f


def _trunk_activations(net, n):
    """The four ReLU outputs of the NatureCNN trunk as float32 NHWC vectors, from whichever buffers the forward pass left: the
    round-3 kernels' float32 activations, or the pre-split ones of the h2 block (h2path.py) unpacked."""
    from srl_amd import hip
    P = "obs_modules_dict.obs."
    bufs = net.ws._bufs
    H2 = f"a:h2[{P}1._Convolution__model.0]."  # the block's own name in front of its buffers (h2path.H2Cnn.pfx), under the pass tag
    if H2 + "a1" not in bufs:
        return [bufs[f"a:{P}1._Convolution__model.{idx}.y"] for idx in (0, 2, 4)] + [bufs[f"a:{P}1._Convolution__model.7.0.y"]]
    from srl_amd.algorithm import h2path
    slots = bufs[H2 + "slots"]
    sp = lambda i: slots.data_ptr() + 4 * i
    a1 = torch.empty(n * 400 * 32, device="cuda:0")
    hip.h2_unpack_rows(bufs[H2 + "a1"].data_ptr(), n * 400, 32, sp(h2path.S_A1), a1.data_ptr(), 32)
    # rows of a sample are in parity-class order: entry = ((y & 1) * 2 + (x & 1)) * 100 + (y >> 1) * 10 + (x >> 1)
    yy, xx = np.meshgrid(np.arange(20), np.arange(20), indexing="ij")
    ent = torch.from_numpy((((yy & 1) * 2 + (xx & 1)) * 100 + (yy >> 1) * 10 + (xx >> 1)).reshape(-1)).to("cuda:0")
    a1 = a1.view(n, 400, 32)[:, ent, :].reshape(-1)
    a2 = torch.empty(n * 81 * 64, device="cuda:0")
    hip.h2_unpack_image(bufs[H2 + "a2"].data_ptr(), n, 9, 9, 64, 0, sp(h2path.S_A2), a2.data_ptr())
    a3 = torch.empty(n * 49 * 64, device="cuda:0")
    hip.h2_unpack_rows(bufs[H2 + "a3"].data_ptr(), n * 49, 64, sp(h2path.S_A3), a3.data_ptr(), 64)
    return [a1, a2, a3, bufs[f"a:{P}1._Convolution__model.7.0.y"]]


def _relu_flips(net, state64, frames):
    """How many ReLU units of the NatureCNN trunk got a different sign on the device than in a float64 forward pass from the
    same parameters.  A pre-activation within float32 rounding of zero flips its gradient mask -- in the reference's own
    float32 arithmetic as easily as here -- and with 256 rows a single flip moves every upstream gradient sum by ~1/256."""
    import torch.nn.functional as F
    P = "obs_modules_dict.obs."
    x = F.layer_norm(torch.from_numpy(frames).double(), (4, 84, 84), state64[P + "0.weight"], state64[P + "0.bias"])
    acts = _trunk_activations(net, frames.shape[0])
    flips = 0
    for li, (idx, stride) in enumerate(((0, 4), (2, 2), (4, 1))):
        w = state64[f"{P}1._Convolution__model.{idx}.weight"]
        x = F.relu(F.conv2d(x, w, state64[f"{P}1._Convolution__model.{idx}.bias"], stride=stride))
        ref = x.permute(0, 2, 3, 1).reshape(-1)  # the device keeps activations NHWC
        got = acts[li][:ref.numel()].cpu()
        flips += int(((got > 0) != (ref > 0)).sum())
        # the activations themselves, against float64 (the pre-split ones carry every element to ~2^-22 of the tensor's bound)
        assert float((got.double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-6), (idx, float((got.double() - ref).abs().max()))
    x = F.relu(F.linear(x.flatten(1), state64[P + "1._Convolution__model.7.0.weight"], state64[P + "1._Convolution__model.7.0.bias"]))
    got = acts[3][:x.numel()].cpu()
    return flips + int(((got > 0) != (x.reshape(-1) > 0)).sum())


@pytest.mark.parametrize("frames", ["noise", "pong"])
@pytest.mark.parametrize("kernels", ["h2", "bf16x3", "f32"])
def test_cnn_step_at_the_benchmarked_dispatch(kernels, frames, monkeypatch):
    """The composition bench.py times, end to end against the oracle: NatureCNN `trainer.step` x 2 with 256 rows in ONE
    row chunk, so that every contraction is above the size gates of gemm.hip / conv.hip and runs on `gemm3_kernel` /
    `obs_*_bf16_kernel` (asserted through the launch counters of the C ABI) -- and the same sample through the float32
    MFMA kernels (SRL_MFMA=f32, SRL_OBS_BF16=0): both kernel sets meet the same oracle at the same tolerances.
    Reference: mappo.py:219-328 (the step), modules/cnn.py:93-135 (the encoder).

    GAE returns and the loss terms: 1e-5 relative against the float32 oracle (the north star's bar).  Gradients: against a
    FLOAT64 run of the same oracle, per parameter tensor no further off (rms) than 3x what the float32 oracle -- the
    reference's arithmetic -- is off itself, with a floor of 1e-5 of the tensor's rms gradient; on `pong`-type frames (flat
    background: the whole-observation LayerNorm cancels catastrophically in float32) the reference is off by up to 1e-2 and
    the device path, which centres the bytes exactly, by 1e-5.  Parameters after each step: 2e-5 absolute, plus, per
    element, what Adam makes of the gradient uncertainty of that element (lr * uncertainty / (sqrt(v) + eps): an element whose
    gradient is below the rounding noise is moved by the SIGN of that noise, in the reference as much as here); both sides
    start every step from the same parameters (the optimiser moments are each side's own)."""
    from srl_amd import hip
    from srl_amd.algorithm import h2path
    # "h2": the default since round 4 -- the convolution stack and its Linear on pre-split activations (h2path.py: csrc/h2conv.h,
    # csrc/h2gemm.h); "bf16x3": round 3's kernels (SRL_H2=0); "f32": the float32 MFMA kernels
    monkeypatch.setattr(h2path, "ENABLED", kernels == "h2")
    if kernels == "f32":
        monkeypatch.setenv("SRL_MFMA", "f32")
        monkeypatch.setenv("SRL_OBS_BF16", "0")
    T, B = 16, 16  # 256 loss rows, one chunk (chunk_rows 16384)
    lr = ATARI_TRAINER["optimizer_config"]["lr"]
    trainer = make_trainer(CNN_POLICY, dict(ATARI_TRAINER))
    net = trainer.policy.net
    oracles = {}
    for dt in (torch.float32, torch.float64):
        onet = OracleActorCritic(**CNN_POLICY, dtype=dt)
        onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
        oracles[dt] = (onet, OracleMappo(onet, **ATARI_TRAINER))
    hip.dispatch_counts(reset=True)
    hip.dispatch_tiles(reset=True)
    tight = total = 0
    carried = {}  # gradient uncertainty of the steps so far: Adam's first moment carries it into the later updates
    for step in range(2):
        arrays = synthetic.make_sample_arrays(seed=71 + step, T=T, B=B, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.05)
        if frames == "pong":
            arrays["obs.obs"] = _pong_frames(np.random.default_rng(step), T + 1, B)
        before = trainer.policy.get_checkpoint()["state_dict"]
        for onet, _ in oracles.values():  # every step starts from the device's parameters on all sides
            for k, p in onet.params.items():
                p.data.copy_(before[k].to(p.dtype))
        sample = synthetic.to_sample_batch(arrays)
        res = trainer.step(sample)
        (onet, oracle), (onet64, oracle64) = oracles[torch.float32], oracles[torch.float64]
        ostats, oout = oracle.step(arrays)
        oracle64.step(arrays)
        assert close(sample.analyzed_result.ret, oout["ret"], 1e-5), step
        assert close(sample.analyzed_result.adv, oout["adv"], 1e-5), step
        for k in ("policy_loss", "value_loss", "entropy"):
            assert abs(res.stats[k] - ostats[k]) <= 1e-5 * max(abs(ostats[k]), 1e-2), (step, k, res.stats[k], ostats[k])
        assert abs(res.stats["grad_norm"] - ostats["grad_norm"]) <= 2e-5 * max(abs(ostats["grad_norm"]), 1e-2), step
        flips = _relu_flips(net, {k: v.double() for k, v in before.items()}, arrays["obs.obs"][:T].reshape(T * B, 4, 84, 84))
        assert flips <= 8, flips  # of 6.6 M units (flat pong frames repeat a near-zero pre-activation at every background position)
        g_hip = net.flat_to_reference(net.grad.detach().cpu())
        after = trainer.policy.get_checkpoint()["state_dict"]
        b2 = oracle.optimizer.param_groups[0]["betas"][1]
        for k, p in onet.params.items():
            g64 = onet64.params[k].grad.numpy()
            e_ref = p.grad.double().numpy() - g64
            e_hip = g_hip[k].double().numpy() - g64
            rms = np.sqrt((g64**2).mean())
            # a flipped ReLU mask moves the sums upstream of it by about one row's share
            bound = max(3.0 * np.sqrt((e_ref**2).mean()), 1e-5 * rms) + flips * 8e-3 * rms
            assert np.sqrt((e_hip**2).mean()) <= bound, (step, k, float(np.sqrt((e_hip**2).mean()) / rms), float(bound / rms), flips)
            st = oracle.optimizer.state[p]
            vhat = (st["exp_avg_sq"] / (1.0 - b2**float(st["step"]))).numpy().astype(np.float64)
            unsure = carried[k] = carried.get(k, 0.0) + 3.0 * (np.abs(e_ref) + np.abs(e_hip)) + 1e-5 * rms
            tol = 2e-5 + lr * np.minimum(2.0, unsure / (np.sqrt(vhat) + 1e-8))
            err = np.abs(after[k].double().numpy() - p.detach().double().numpy())
            assert (err <= tol).all(), (step, k, float(err.max()), float((err / tol).max()))
            tight += int((tol <= 3e-5).sum())
            total += err.size
    assert tight >= 0.9 * total, (tight, total)  # the widened tolerances are the exception
    counts = hip.dispatch_counts(reset=True)
    tiles = hip.dispatch_tiles(reset=True)
    if kernels == "h2":
        from conftest import BENCH_CHUNK_TILES, tile_kinds
        assert tile_kinds(tiles) == tile_kinds(BENCH_CHUNK_TILES), tiles  # the instantiations a 16 384-row chunk of the benchmark runs
        # per step: conv2, conv3, Linear forward; Linear and both convolutions' data gradients; the three weight gradients (round 6:
        # the Linear's on csrc/h2tn.h) -- 9 launches of the pre-split family; the first layer on the byte kernels; nothing on the
        # float32 MFMA kernels or round 3's
        assert counts["h2"] == 2 * 9 and counts["gemm_f32"] == 0 and counts["gemm2h"] + counts["gemm3"] == 0, counts
        assert counts["obs_fwd_bf16"] == 2 and counts["obs_bwd_bf16"] == 2, counts
    elif kernels == "bf16x3":
        # conv2/conv3 forward, their weight and data gradients, the three FC products: all on the bf16 matrix cores,
        # the first layer on the byte kernels, only the two heads (N = 6, N = 1) on the skinny kernels
        # (every one of them knows both operands' ranges -- tracked by the producing kernels -- and runs on two f16 pieces
        # per operand, `gemm2h`: conv2 / conv3 / FC x forward, weight gradient, data gradient)
        import os
        if os.environ.get("SRL_F16X2", "1")[:1] != "0":  # (that A/B switch sends them back to three bf16 pieces: `gemm3`)
            assert counts["gemm_f32"] == 0 and counts["gemm3"] == 0, counts
            assert counts["gemm2h"] == 2 * 9, counts
        assert counts["obs_fwd_bf16"] == 2 and counts["obs_bwd_bf16"] == 2, counts
    else:
        assert counts["gemm3"] == 0 and counts["gemm2h"] == 0 and counts["obs_fwd_bf16"] == 0 and counts["obs_bwd_bf16"] == 0, counts
        assert counts["gemm_f32"] > 0, counts


def test_checkpoint_roundtrip_and_reuse():
    trainer = make_trainer(C1_POLICY, dict(popart=False, optimizer_config=dict(lr=1e-3), recompute_adv_on_reuse=False))
    arrays = synthetic.make_sample_arrays(seed=1, T=16, B=4, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2)
    sample = synthetic.to_sample_batch(arrays)
    trainer.step(sample)
    adv_first = sample.analyzed_result.adv.copy()
    ckpt = copy.deepcopy(trainer.get_checkpoint())
    assert set(ckpt) == {"steps", "state_dict", "optimizer_state_dict"} and ckpt["steps"] == 0
    # torch's own Adam accepts the optimiser state (same structure as the reference's checkpoints)
    params = [torch.nn.Parameter(v.clone()) for v in ckpt["state_dict"].values()]
    torch.optim.Adam(params, lr=1e-3).load_state_dict(ckpt["optimizer_state_dict"])
    r1 = trainer.step(sample)  # re-use: advantages in the sample are kept (recompute_adv_on_reuse=False)
    assert np.array_equal(sample.analyzed_result.adv, adv_first)
    other = make_trainer(dict(C1_POLICY, seed=99), dict(popart=False, optimizer_config=dict(lr=1e-3),
                                                        recompute_adv_on_reuse=False))
    other.load_checkpoint(ckpt)
    assert other.policy.version == 0
    r2 = other.step(sample)
    for k in ("policy_loss", "value_loss", "grad_norm"):
        assert abs(r1.stats[k] - r2.stats[k]) <= 1e-6 * max(abs(r1.stats[k]), 1e-3), k


def test_device_resident_sample_matches_host_sample():
    a = make_trainer(C1_POLICY, dict(popart=False))
    b = make_trainer(C1_POLICY, dict(popart=False))
    arrays = synthetic.make_sample_arrays(seed=2, T=16, B=4, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2)
    host = synthetic.to_sample_batch(arrays)
    devs = synthetic.to_sample_batch({k: torch.from_numpy(v).to("cuda:0") for k, v in arrays.items()})
    ra, rb = a.step(host), b.step(devs)
    for k in ("policy_loss", "value_loss"):  # the same forward pass
        assert ra.stats[k] == rb.stats[k]
    # gradients: the fused small-MLP backward adds its 16-row partial sums with float atomics (order not fixed run to run)
    assert abs(ra.stats["grad_norm"] - rb.stats["grad_norm"]) <= 1e-6 * abs(ra.stats["grad_norm"])
    assert isinstance(devs.analyzed_result.adv, torch.Tensor)


def test_implicit_and_explicit_conv_paths_agree():
    """The implicit-GEMM convolution path and the im2col fallback are two implementations of the same step."""
    arrays = synthetic.make_sample_arrays(seed=3, T=6, B=5, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.1)
    stats = []
    import os
    for explicit in (False, True):
        os.environ["SRL_EXPLICIT_CONV"] = "1" if explicit else "0"
        try:
            tr = make_trainer(CNN_POLICY, ATARI_TRAINER)
        finally:
            os.environ.pop("SRL_EXPLICIT_CONV")
        assert tr.policy.net.force_explicit_conv == explicit
        res = tr.step(synthetic.to_sample_batch(arrays))
        stats.append((res.stats, tr.policy.get_checkpoint()["state_dict"]))
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        a, b = stats[0][0][k], stats[1][0][k]
        assert abs(a - b) <= 1e-5 * max(abs(b), 1e-2), (k, a, b)
    # one Adam step moves a weight by ~lr = 5e-4 times g / (|g| + eps): where a gradient is within rounding of zero the two
    # implementations (different summation orders) may step it differently by a fraction of lr
    for k in stats[0][1]:
        assert np.abs(stats[0][1][k].numpy() - stats[1][1][k].numpy()).max() <= 5e-5, k


def test_batcher_device_staging_matches_host_concatenation():
    """The batcher's device block (no host np.concatenate of the observations) feeds rollout the same rows."""
    from srl_amd.namedarray import NamedArray
    from srl_amd.runtime.batcher import InferenceBatcher
    pol = policy_api.make(config.Policy("actor-critic", args=dict(C1_POLICY, seed=11)))
    rng = np.random.default_rng(3)

    def req(ids):
        n = len(ids)
        return policy_api.RolloutRequest(obs=NamedArray(obs=rng.standard_normal((n, 4)).astype(np.float32)),
                                         is_evaluation=np.ones((n, 1), np.uint8), on_reset=np.zeros((n, 1), np.uint8),
                                         client_id=np.array(ids, np.int32).reshape(n, 1),
                                         request_id=np.arange(n).reshape(n, 1), received_time=np.zeros((n, 1), np.int64),
                                         buffer_index=np.zeros((n, 1), np.int32))

    reqs = [req([0, 1, 2]), req([3]), req([4, 5, 6, 7, 8])]
    outs = []
    for staged in (True, False):
        b = InferenceBatcher(pol, batch_size=7, stage_on_device=staged)
        for r in reqs:
            b.post(r)
        res = b.poll()
        assert [o.action.x.shape[0] for o in res] == [7, 2]
        outs.append(res)
    for a, c in zip(*outs):
        assert np.array_equal(a.client_id, c.client_id) and np.array_equal(a.action.x, c.action.x)
        assert np.array_equal(a.analyzed_result.value, c.analyzed_result.value)
        assert np.array_equal(a.analyzed_result.log_probs, c.analyzed_result.log_probs)


def test_recurrent_rollout_golden(golden):
    """Stateful rollout: [n, layers, H] states in, new states out, against the reference (deterministic actions)."""
    g = golden("steps_rnn.npz")
    pol = policy_api.make(config.Policy("actor-critic", args=CASES["gru"][0]))
    pol.load_checkpoint({"steps": 0, "state_dict": {k[len("roll_param:"):]: torch.from_numpy(g[k]) for k in g.files
                                                    if k.startswith("roll_param:")}})
    assert pol.default_policy_state.hx.shape == (1, 32)
    n = g["roll_obs"].shape[0]
    req = policy_api.RolloutRequest(obs=NamedArray(obs=g["roll_obs"]), policy_state=NamedArray(hx=g["roll_hx"]),
                                    is_evaluation=np.ones((n, 1), np.uint8), on_reset=np.zeros((n, 1), np.uint8))
    res = pol.rollout(req)
    assert np.array_equal(res.action.x, g["roll_action"])
    assert close(res.analyzed_result.log_probs, g["roll_log_probs"], 1e-5)
    assert close(res.analyzed_result.value, g["roll_value"], 1e-5)
    assert res.policy_state.hx.shape == (n, 1, 32) and close(res.policy_state.hx, g["roll_new_hx"], 1e-5)


def test_continuous_rollout_golden(golden):
    """Gaussian head: evaluation rollout returns the mean and its log-probability; sampling is checked statistically."""
    g = golden("steps_continuous.npz")
    pol = policy_api.make(config.Policy("gym_mujoco", args=dict(_CBASE, std_type="separate_learnable", seed=32)))
    pol.load_checkpoint({"steps": 0, "state_dict": {k[len("roll_param:"):]: torch.from_numpy(g[k]) for k in g.files
                                                    if k.startswith("roll_param:")}})
    n = g["roll_obs"].shape[0]
    req = policy_api.RolloutRequest(obs=NamedArray(obs=g["roll_obs"]), is_evaluation=np.ones((n, 1), np.uint8),
                                    on_reset=np.zeros((n, 1), np.uint8))
    res = pol.rollout(req)
    assert res.action.x.dtype == np.float32 and close(res.action.x, g["roll_action"], 1e-5)
    assert close(res.analyzed_result.log_probs, g["roll_log_probs"], 1e-5)
    assert close(res.analyzed_result.value, g["roll_value"], 1e-5)
    # stochastic: (x - mean) / std is standard normal; the reported log-prob is that of the returned action
    m = 4096
    obs = np.repeat(g["roll_obs"][:1], m, 0)
    req = policy_api.RolloutRequest(obs=NamedArray(obs=obs), is_evaluation=np.zeros((m, 1), np.uint8),
                                    on_reset=np.zeros((m, 1), np.uint8))
    s = pol.rollout(req)
    std = np.exp(g["roll_param:log_std"])
    z = (s.action.x - g["roll_action"][:1]) / std
    assert abs(z.mean()) < 0.05 and abs(z.std() - 1.0) < 0.05
    lp = (-0.5 * z**2 - np.log(std) - 0.5 * np.log(2 * np.pi)).sum(-1, keepdims=True)
    assert close(s.analyzed_result.log_probs, lp, 1e-4)


@pytest.mark.parametrize("T,B", [(2, 1), (5, 7), (1, 3), (33, 2)])
def test_ragged_and_tiny_batches_vs_oracle(T, B):
    """Shapes that hit none of the vectorised fast paths (B not a multiple of 4, a single column, a single rewarding
    row): the generic scan / loss / GEMM paths against the CPU oracle."""
    pargs = dict(C1_POLICY, layernorm=True, popart=True, seed=4)
    targs = dict(popart=True, optimizer_config=dict(lr=1e-3), max_grad_norm=1.0)
    trainer = make_trainer(pargs, targs)
    onet = OracleActorCritic(**pargs)
    onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **targs)
    arrays = synthetic.make_sample_arrays(seed=T * 10 + B, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2,
                                          p_done=0.2)
    arrays["on_reset"][1:] = 0  # keep every step in the loss mask: with so few steps an all-masked batch (0/0 in the
    arrays["done"][:] = 0       # reference too) would otherwise be likely
    arrays["truncated"][:] = 0
    sample = synthetic.to_sample_batch(arrays)
    res = trainer.step(sample)
    ostats, oout = oracle.step(arrays)
    assert close(sample.analyzed_result.ret, oout["ret"], 1e-5) and close(sample.analyzed_result.adv, oout["adv"], 1e-5)
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (k, res.stats[k], ostats[k])
    # a single rollout request, and an empty one
    pol = trainer.policy
    one = pol.rollout(policy_api.RolloutRequest(obs=NamedArray(obs=np.zeros((1, 4), np.float32)),
                                                is_evaluation=np.ones((1, 1), np.uint8), on_reset=np.zeros((1, 1), np.uint8)))
    assert one.action.x.shape == (1, 1) and np.isfinite(one.analyzed_result.value).all()


def test_streamed_rollout_matches_one_piece():
    """Big host batches go through rollout in pieces with the copies overlapped; outputs must not depend on the split."""
    pol = policy_api.make(config.Policy("actor-critic", args=CNN_POLICY))
    n = 2 * pol.ROLLOUT_PIECE + 300
    rng = np.random.default_rng(3)
    obs = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    req = policy_api.RolloutRequest(obs=NamedArray(obs=obs), is_evaluation=np.ones((n, 1), np.uint8),
                                    on_reset=np.zeros((n, 1), np.uint8))
    streamed = pol.rollout(req)
    pol.ROLLOUT_PIECE = n  # one piece: the plain path
    whole = pol.rollout(req)
    assert np.array_equal(streamed.action.x, whole.action.x)
    assert close(streamed.analyzed_result.log_probs, whole.analyzed_result.log_probs, 1e-6)
    assert close(streamed.analyzed_result.value, whole.analyzed_result.value, 1e-6)
    # sampling mode: same distribution family, reproducible for the same policy state
    req_s = policy_api.RolloutRequest(obs=NamedArray(obs=obs[:64]), is_evaluation=np.zeros((64, 1), np.uint8),
                                      on_reset=np.zeros((64, 1), np.uint8))
    assert pol.rollout(req_s).action.x.shape == (64, 1)


def test_served_batches_keep_derived_weights_until_the_parameters_change():
    """Between the request batches a policy serves, what its executor derived from the parameters (the first layer's folded
    weights, the pre-split weight copies of the h2 path) is kept -- and dropped when the parameters change through
    `load_checkpoint` (a parameter pull): the next batch must be what a policy built from those parameters gives."""
    n = 4096   # the h2 path's row count
    rng = np.random.default_rng(17)
    obs = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    req = policy_api.RolloutRequest(obs=NamedArray(obs=obs), is_evaluation=np.ones((n, 1), np.uint8), on_reset=np.zeros((n, 1), np.uint8))
    pol = policy_api.make(config.Policy("actor-critic", args=CNN_POLICY))
    other = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=CNN_POLICY.get("seed", 1) + 5)))
    from srl_amd import hip
    hip.dispatch_counts(reset=True)
    first = pol.rollout(req)
    launches_first = sum(hip.dispatch_counts(reset=True).values())
    again = pol.rollout(req)
    launches_again = sum(hip.dispatch_counts(reset=True).values())
    assert launches_again == launches_first   # (the derived-weight kernels are not in the counted families; the products are)
    assert np.array_equal(first.action.x, again.action.x) and np.array_equal(first.analyzed_result.value, again.analyzed_result.value)
    want = other.rollout(req)
    assert not np.array_equal(want.analyzed_result.value, first.analyzed_result.value)
    pol.load_checkpoint(other.get_checkpoint())   # (api/policy.py's parameter pull)
    got = pol.rollout(req)
    assert np.array_equal(got.action.x, want.action.x)
    assert np.array_equal(got.analyzed_result.value, want.analyzed_result.value)
    assert np.array_equal(got.analyzed_result.log_probs, want.analyzed_result.log_probs)


def test_streamed_rollout_with_action_mask_matches_one_piece():
    """The availability mask of a streamed piece is staged on the side stream like the frames: a piece must sample
    with ITS OWN mask (every action legal under it) and give the one-piece result."""
    pol = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=8)))
    pol.ROLLOUT_PIECE = 256
    n = 6 * pol.ROLLOUT_PIECE + 77
    rng = np.random.default_rng(11)
    obs = rng.integers(0, 256, size=(n, 4, 84, 84), dtype=np.uint8)
    A = CNN_POLICY["action_dim"]
    avail = np.zeros((n, A), dtype=np.uint8)
    avail[np.arange(n), rng.integers(0, A, size=n)] = 1  # exactly one legal action per row, different between pieces
    for mode in (1, 0):  # evaluation (argmax) and sampling: with one legal action both are fully determined
        req = policy_api.RolloutRequest(obs=NamedArray(obs=obs, available_action=avail),
                                        is_evaluation=np.full((n, 1), mode, np.uint8), on_reset=np.zeros((n, 1), np.uint8))
        for _ in range(3):
            out = pol.rollout(req)
            assert np.array_equal(out.action.x[:, 0], avail.argmax(1)), "a piece used another piece's mask"
            assert np.abs(out.analyzed_result.log_probs).max() < 1e-4  # log-prob of the only legal action ~ 0
    pol.ROLLOUT_PIECE = n
    whole = pol.rollout(req)
    assert np.array_equal(whole.action.x, out.action.x)
    assert close(whole.analyzed_result.value, out.analyzed_result.value, 1e-6)


def test_float32_frames_and_mixed_observation_keys_vs_oracle():
    """Image observations delivered as float32 (the non-byte staging of the first layer), next to a vector key and an
    action mask, separate backbones: one trainer step against the CPU oracle."""
    pargs = dict(obs_dim={"img": (4, 20, 20), "vec": 6}, action_dim=5, hidden_dim=32, num_dense_layers=1, num_rnn_layers=0,
                 popart=False, layernorm=True, shared_backbone=False, chunk_len=4, seed=12,
                 cnn_layers=dict(img=[(8, 4, 4, 0, 'zeros'), (16, 3, 1, 0, 'zeros')]))
    targs = dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=5.0, chunk_rows=50)
    trainer = make_trainer(pargs, targs)
    onet = OracleActorCritic(**pargs)
    onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **{k: v for k, v in targs.items() if k != "chunk_rows"})
    arrays = synthetic.make_sample_arrays(seed=21, T=10, B=9, obs_spec={"img": ((4, 20, 20), "f32"), "vec": ((6,), "f32")},
                                          action_dims=5, p_done=0.1, available_action=True)
    sample = synthetic.to_sample_batch(arrays)
    res = trainer.step(sample)
    ostats, oout = oracle.step(arrays)
    assert close(sample.analyzed_result.ret, oout["ret"], 1e-5)
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (k, res.stats[k], ostats[k])
    sd, osd = trainer.policy.get_checkpoint()["state_dict"], onet.state_dict()
    for k in sd:
        d = np.abs(sd[k].numpy() - osd[k].numpy())
        assert d.max() <= 1e-3 and np.mean(d > 1e-6) < 2e-3, (k, d.max())


def test_sample_of_another_shape_is_refused_not_read():
    """The kernels take sizes from the network's spec: observations of another shape must raise, not fault."""
    from srl_amd import hip
    tr = make_trainer(CNN_POLICY, dict(ATARI_TRAINER))
    arrays = synthetic.make_sample_arrays(seed=1, T=3, B=2, obs_spec={"obs": ((4, 42, 42), "u8")}, action_dims=6)
    with pytest.raises(hip.HipError, match="image observation"):
        tr.step(synthetic.to_sample_batch(arrays))
    tv = make_trainer(C1_POLICY, dict(popart=False))
    arrays = synthetic.make_sample_arrays(seed=1, T=3, B=2, obs_spec={"obs": ((6,), "f32")}, action_dims=2)
    with pytest.raises(hip.HipError, match="vector observation|chain"):
        tv.step(synthetic.to_sample_batch(arrays))
