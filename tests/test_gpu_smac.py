"""GPU parity of the multi-agent path (BASELINE config 4: SMAC 3m, shared agents): ``smac_rnn`` policy + ``mappo`` trainer on
``[Tb, B, agents, ...]`` samples against golden vectors from the real reference (tests/golden/gen_golden.py gen_smac)."""
import numpy as np
import pytest
import torch

import srl_amd
from srl_amd.api import config, policy as policy_api, trainer as trainer_api
from srl_amd.namedarray import NamedArray
from srl_amd.runtime import synthetic

srl_amd.register_all()
pytestmark = pytest.mark.gpu

H, A, CL = 32, 3, 5
POLICY = dict(map_name="3m", hidden_dim=H, chunk_len=CL, seed=31, shared=True)
TRAINER = dict(popart=True, ppo_epochs=2, optimizer_config=dict(lr=5e-4, eps=1e-5), max_grad_norm=10.0,
               value_loss="huber", value_loss_config=dict(delta=10.0), clip_value=True, dual_clip=False)
SAMPLE = dict(T=20, B=4, agents=A, obs_spec={"local_obs": ((30,), "f32"), "state": ((48,), "f32")}, action_dim=9,
              p_done=0.08, policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})


def close(a, b, rtol, scale=1.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool((np.abs(a - b) <= rtol * np.maximum(np.abs(b), scale)).all())


def test_smac_steps_match_reference_golden(golden):
    g = golden("steps_smac.npz")
    trainer = trainer_api.make(config.Trainer("mappo", args=TRAINER), config.Policy("smac_rnn", args=POLICY))
    for k, v in trainer.policy.get_checkpoint()["state_dict"].items():  # same seed -> the reference's initial weights
        assert np.allclose(v.numpy(), g[f"smac_init_param:{k}"], rtol=1e-4, atol=1e-4), k
    names = list(g["smac_stat_names"])
    for step in range(2):
        arrays = synthetic.make_multiagent_arrays(seed=300 + step, **SAMPLE)
        sample = synthetic.to_sample_batch(arrays)
        if step == 0:
            Tb = arrays["on_reset"].shape[0]
            ar = trainer.policy.analyze(sample[:Tb - 1], target="ppo")
            lp, ref_lp = ar.new_action_log_probs.cpu().numpy(), g["smac_analyze_new_lp"]
            assert lp.shape == ref_lp.shape == (Tb - 1, 4, A, 1)
            dead = arrays["obs.is_alive"][:Tb - 1] == 0
            assert np.array_equal(np.isneginf(lp), dead) and np.array_equal(np.isneginf(ref_lp), dead)
            assert close(lp[~dead], ref_lp[~dead], 1e-5), "analyze log-probs"
            assert close(ar.state_values.cpu().numpy(), g["smac_analyze_value"], 1e-5), "analyze values"
            assert close(ar.entropy.cpu().numpy(), g["smac_analyze_entropy"], 1e-5), "analyze entropy"
        res = trainer.step(sample)
        ref = dict(zip(names, g[f"smac_step{step}_stats"]))
        for k in ("policy_loss", "value_loss", "entropy", "advantage", "value_targets", "importance_weight", "clip_ratio",
                  "done", "truncated", "grad_norm", "frames", "denorm_value"):
            tol = 1e-5 if k in ("policy_loss", "value_loss", "entropy", "value_targets", "denorm_value") else 1e-4
            assert abs(res.stats[k] - ref[k]) <= tol * max(abs(ref[k]), 1e-2), (step, k, res.stats[k], ref[k])
        if step == 0:
            assert sample.analyzed_result.adv.shape == g["smac_step0_adv"].shape  # [Tb, B, agents, 1]
            assert close(sample.analyzed_result.adv, g["smac_step0_adv"], 1e-5)
            assert close(sample.analyzed_result.ret, g["smac_step0_ret"], 1e-5)
        sd = trainer.policy.get_checkpoint()["state_dict"]
        pre = f"smac_step{step}_param:"
        for key in g.files:
            if key.startswith(pre):
                got = sd[key[len(pre):]].numpy()
                assert np.abs(got - g[key]).max() <= 2e-5, (step, key, np.abs(got - g[key]).max())
                if "_RunningMeanStd__" in key:
                    assert got.dtype == np.float64 and np.allclose(got, g[key], rtol=1e-6, atol=1e-13), (step, key)
    assert trainer.policy.version == int(g["smac_version"]) and res.step == trainer.policy.version


def test_smac_rollout_golden(golden):
    """[N, agents, ...] requests: agents folded for the network, carried LSTM states zeroed where the episode restarts."""
    g = golden("steps_smac.npz")
    pol = policy_api.make(config.Policy("smac_rnn", args=POLICY))
    pol.load_checkpoint({"steps": 2, "state_dict": {k[len("smac_step1_param:"):]: torch.from_numpy(g[k]) for k in g.files
                                                    if k.startswith("smac_step1_param:")}})
    assert pol.default_policy_state.actor_hx.shape == (A, 1, 2 * H)
    N = g["smac_roll_on_reset"].shape[0]
    obs = NamedArray(**{k[len("smac_roll_obs."):]: g[k] for k in g.files if k.startswith("smac_roll_obs.")})
    req = policy_api.RolloutRequest(obs=obs, policy_state=NamedArray(actor_hx=g["smac_roll_actor_hx"],
                                                                     critic_hx=g["smac_roll_critic_hx"]),
                                    is_evaluation=np.ones((N, A, 1), np.uint8), on_reset=g["smac_roll_on_reset"])
    res = pol.rollout(req)
    assert res.action.x.shape == (N, A, 1) and np.array_equal(res.action.x, g["smac_roll_action"])
    assert close(res.analyzed_result.log_probs, g["smac_roll_log_probs"], 1e-5)
    assert close(res.analyzed_result.value, g["smac_roll_value"], 1e-5)
    assert close(res.policy_state.actor_hx, g["smac_roll_new_actor_hx"], 1e-5)
    assert close(res.policy_state.critic_hx, g["smac_roll_new_critic_hx"], 1e-5)
    # one evaluation flag per environment is accepted too, and no carried state means the default (zero) state
    res2 = pol.rollout(policy_api.RolloutRequest(obs=obs, policy_state=None, is_evaluation=np.ones((N, 1), np.uint8),
                                                 on_reset=np.ones((N, A, 1), np.uint8)))
    assert res2.action.x.shape == (N, A, 1) and res2.policy_state.actor_hx.shape == (N, A, 1, 2 * H)


def test_smac_non_shared_agents_are_plain_columns():
    """shared=False: every agent is its own stream, samples are [Tb, B, ...]; same network, no agent axis."""
    pol_args = dict(POLICY, shared=False)
    trainer = trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, ppo_epochs=1)), config.Policy("smac_rnn", args=pol_args))
    assert trainer.policy.default_policy_state.critic_hx.shape == (1, 2 * H)
    flat = synthetic.make_multiagent_arrays(seed=7, **SAMPLE)
    flat.pop("obs.is_alive")  # not part of a non-shared environment's observation (smac_env.py:243-247)
    arrays = {k: v.reshape(v.shape[0], v.shape[1] * v.shape[2], *v.shape[3:]) for k, v in flat.items()}
    shared = trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, ppo_epochs=1)), config.Policy("smac_rnn", args=POLICY))
    r1 = trainer.step(synthetic.to_sample_batch(arrays))
    r2 = shared.step(synthetic.to_sample_batch(flat))
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        assert abs(r1.stats[k] - r2.stats[k]) <= 1e-6 * max(1.0, abs(r2.stats[k])), k


def test_football_smm_preset_downsized_vs_oracle():
    """The football-smm preset (default convolution stack of modules/cnn.py:96-98 with its halving Linear tower, GRU,
    PopArt, separate backbones) on down-sized frames: a trainer step against the CPU oracle.  (At the preset's
    (4, 96, 72) frames the first Linear alone has 254 M weights; the arithmetic is the same.)"""
    from oracle.net import OracleActorCritic
    from oracle.trainer import OracleMappo
    shape = (4, 24, 20)
    pargs = dict(obs_dim={"obs": shape}, hidden_dim=32, seed=3, chunk_len=4)
    targs = dict(popart=True, optimizer_config=dict(lr=5e-4), max_grad_norm=10.0)
    trainer = trainer_api.make(config.Trainer("mappo", args=targs), config.Policy("football-smm-separate", args=pargs))
    spec = trainer.policy.spec
    assert spec.num_rnn_layers == 1 and spec.popart and not spec.shared_backbone and spec.act_dims == [19]
    tower = [L for L in spec.obs_encoders[0].layers if type(L).__name__ == "LinearSpec"]
    assert [(L.in_features, L.out_features) for L in tower] == [(768, 384), (384, 192), (192, 32)]  # halved until <= 8 H
    oargs = dict(obs_dim={"obs": shape}, action_dim=19, hidden_dim=32, num_rnn_layers=1, rnn_type="gru", popart=True,
                 chunk_len=4, shared_backbone=False,
                 cnn_layers={"obs": [(4, 5, 1, 0, "zeros"), (8, 3, 1, 0, "zeros"), (4, 3, 1, 0, "zeros")]})
    onet = OracleActorCritic(**oargs)
    onet.load_state_dict({k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **targs)
    arrays = synthetic.make_sample_arrays(seed=5, T=8, B=6, obs_spec={"obs": (shape, "u8")}, action_dims=19, p_done=0.1,
                                          policy_state={"actor_hx": (1, 32), "critic_hx": (1, 32)})
    sample = synthetic.to_sample_batch(arrays)
    res = trainer.step(sample)
    ostats, oout = oracle.step(arrays)
    assert close(sample.analyzed_result.ret, oout["ret"], 1e-5)
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "denorm_value"):
        assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (k, res.stats[k], ostats[k])
    sd, osd = trainer.policy.get_checkpoint()["state_dict"], onet.state_dict()
    for k in sd:  # the first Adam step moves a weight by lr * g / (|g| + eps): where |g| ~ eps, rounding noise in g is
        # amplified to a fraction of lr; everywhere else the two agree to ~1e-8
        d = np.abs(sd[k].numpy() - osd[k].numpy())
        assert d.max() <= 5e-4 and np.mean(d > 1e-6) < 1e-3, (k, d.max(), np.mean(d > 1e-6))


@pytest.mark.parametrize("kind", ["vector", "frames"])
def test_encoder_pieces_do_not_change_the_step(kind):
    """Recurrent nets hand every row of the sample to the network at once; the (row-independent) encoders then run
    in pieces bounded by 32-bit addressing of their activations.  Forcing tiny pieces must give the same update."""
    if kind == "vector":
        name, pargs = "smac_rnn", POLICY
        arrays = synthetic.make_multiagent_arrays(seed=9, **SAMPLE)
    else:
        name, pargs = "football-smm-separate", dict(obs_dim={"obs": (4, 24, 20)}, hidden_dim=32, seed=3, chunk_len=4,
                                                     rnn_type="lstm")
        arrays = synthetic.make_sample_arrays(seed=5, T=8, B=6, obs_spec={"obs": ((4, 24, 20), "u8")}, action_dims=19,
                                              p_done=0.1, policy_state={"actor_hx": (1, 64), "critic_hx": (1, 64)})
    targs = dict(popart=True, optimizer_config=dict(lr=5e-4), max_grad_norm=10.0)
    whole = trainer_api.make(config.Trainer("mappo", args=targs), config.Policy(name, args=pargs))
    pieces = trainer_api.make(config.Trainer("mappo", args=targs), config.Policy(name, args=pargs))
    pieces.policy.net.encoder_rows = 13
    r1 = whole.step(synthetic.to_sample_batch({k: v.copy() for k, v in arrays.items()}))
    r2 = pieces.step(synthetic.to_sample_batch({k: v.copy() for k, v in arrays.items()}))
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        assert abs(r1.stats[k] - r2.stats[k]) <= 1e-6 * max(1.0, abs(r1.stats[k])), (k, r1.stats[k], r2.stats[k])
    d = (whole.policy.net.flat - pieces.policy.net.flat).abs()
    assert float(d.max()) <= 5e-4 and float((d > 1e-6).float().mean()) < 1e-3
    # and through rollout (forward only)
    n = 40
    rng = np.random.default_rng(0)
    if kind == "vector":
        obs = NamedArray(local_obs=rng.standard_normal((n, A, 30)).astype(np.float32),
                         state=rng.standard_normal((n, A, 48)).astype(np.float32),
                         available_action=np.ones((n, A, 9), np.uint8))
        req = policy_api.RolloutRequest(obs=obs, policy_state=None, is_evaluation=np.ones((n, 1), np.uint8),
                                        on_reset=np.ones((n, A, 1), np.uint8))
    else:
        obs = NamedArray(obs=rng.integers(0, 256, (n, 4, 24, 20), dtype=np.uint8))
        z = np.zeros((n, 1, 64), np.float32)
        req = policy_api.RolloutRequest(obs=obs, policy_state=NamedArray(actor_hx=z, critic_hx=z),
                                        is_evaluation=np.ones((n, 1), np.uint8), on_reset=np.zeros((n, 1), np.uint8))
    pieces.policy.load_checkpoint(whole.policy.get_checkpoint())
    a, b = whole.policy.rollout(req), pieces.policy.rollout(req)
    assert np.array_equal(a.action.x, b.action.x) and close(a.analyzed_result.value, b.analyzed_result.value, 1e-6)
