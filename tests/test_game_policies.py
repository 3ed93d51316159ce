"""CPU checks of the per-game policies: parameter tables and initial weights against the reference (fixtures from
tests/golden/gen_golden.py gen_presets / gen_smac), registration names, and the SMAC oracle against the reference."""
import numpy as np
import pytest
import torch

import srl_amd
from oracle.net import OracleSMACNet
from oracle.trainer import OracleMappo
from srl_amd.algorithm import game_policies as gp
from srl_amd.algorithm import netspec as ns
from srl_amd.api import config, policy as policy_api
from srl_amd.runtime import synthetic

srl_amd.register_all()

PRESETS = {"football-simple115-separate": gp.FootballSeparatePolicy, "overcooked-separate": gp.OvercookedSeparatePolicy,
           "atari-vision": gp.AtariVisionPolicy, "football-smm-separate": gp.FootballSMMPolicy}


def test_game_policies_registered_under_reference_names():
    names = set(policy_api.ALL_POLICY_CLASSES)
    assert set(PRESETS) | {"atari_naive_rnn", "smac_rnn", "gym_mujoco"} <= names


@pytest.mark.parametrize("name", list(PRESETS))
def test_preset_param_table_matches_reference(name, golden):
    g = golden("presets.npz")
    args = {k: v for k, v in PRESETS[name].defaults.items() if k != "chunk_len"}
    with_values = name != "football-smm-separate"  # 254 M weights: table only (no initialisation run)
    spec, vals = ns.build_netspec(**dict(args, seed=41 if with_values else None))
    keys = [info.key for info in spec.params.values()] + (list(spec.popart_keys) if spec.popart else [])
    assert keys == list(g[f"{name}:keys"])
    shapes = {k: tuple(int(x) for x in s.split(",") if x) for k, s in zip(g[f"{name}:keys"], g[f"{name}:shapes"])}
    for info in spec.params.values():
        assert info.ref_shape == shapes[info.key], info.key
    if with_values:
        for k, v in vals.items():
            stride = 97 if v.numel() < 100000 else 4999
            assert np.allclose(v.numpy().reshape(-1)[::stride], g[f"{name}:init_s{stride}:{k}"], rtol=1e-4, atol=1e-4), k


def test_smac_param_table_and_init(golden):
    g = golden("steps_smac.npz")
    spec, vals = ns.build_smac_netspec(30, 48, 9, 32, seed=31)
    ref_keys = [k[len("smac_init_param:"):] for k in g.files if k.startswith("smac_init_param:")]
    assert list(vals) == ref_keys  # state_dict order: the optimiser state of a checkpoint is indexed by it
    assert spec.rnn_state_width == 64 and spec.actor_backbone[0].kind == "lstm"  # AutoResetRNN's default cell
    for k, v in vals.items():
        assert v.dtype == (torch.float64 if "_RunningMeanStd__" in k else torch.float32)
        assert np.allclose(v.numpy(), g[f"smac_init_param:{k}"], rtol=1e-4, atol=1e-4), k
    with pytest.raises(NotImplementedError):
        policy_api.make(config.Policy("smac_rnn", args=dict(map_name="3m", agent_specific_obs=True)))
    with pytest.raises(ValueError):
        policy_api.make(config.Policy("smac_rnn", args=dict(map_name="not-a-map")))


def test_smac_oracle_steps_golden(golden):
    """OracleSMACNet + OracleMappo on [Tb, B, agents, ...] samples against the reference's trainer (gen_smac)."""
    g = golden("steps_smac.npz")
    net = OracleSMACNet(30, 48, 9, 32, 5)
    net.load_state_dict({k[len("smac_init_param:"):]: g[k] for k in g.files if k.startswith("smac_init_param:")})
    tr = OracleMappo(net, popart=True, ppo_epochs=2, optimizer_config=dict(lr=5e-4, eps=1e-5), max_grad_norm=10.0,
                     value_loss="huber", value_loss_config=dict(delta=10.0), clip_value=True, dual_clip=False)
    names = list(g["smac_stat_names"])
    for step in range(2):
        arrays = synthetic.make_multiagent_arrays(seed=300 + step, T=20, B=4, agents=3,
                                                  obs_spec={"local_obs": ((30,), "f32"), "state": ((48,), "f32")},
                                                  action_dim=9, p_done=0.08,
                                                  policy_state={"actor_hx": (1, 64), "critic_hx": (1, 64)})
        stats, out = tr.step(arrays)
        ref = dict(zip(names, g[f"smac_step{step}_stats"]))
        for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "clip_ratio", "importance_weight", "denorm_value"):
            assert abs(stats[k] - ref[k]) <= 2e-5 * max(1.0, abs(ref[k])), (step, k)
        if step == 0:
            assert np.array_equal(out["adv"], g["smac_step0_adv"]) and np.array_equal(out["ret"], g["smac_step0_ret"])
    sd = net.state_dict()
    for k in sd:
        assert np.allclose(sd[k].numpy(), g[f"smac_step1_param:{k}"], rtol=1e-4, atol=1e-6), k


def test_multiagent_sample_invariants():
    a = synthetic.make_multiagent_arrays(seed=3, T=12, B=5, agents=3, obs_spec={"local_obs": ((30,), "f32")}, action_dim=9)
    for k in ("done", "truncated", "on_reset"):
        assert a[k].shape == (13, 5, 3, 1) and (a[k] == a[k][:, :, :1]).all()  # per environment, same for its agents
    assert not (a["done"] & a["truncated"]).any()
    assert (a["reward"][:-1][a["on_reset"][1:] == 1] == 0).all() and (a["analyzed_result.value"][a["done"] == 1] == 0).all()
    taken = np.take_along_axis(a["obs.available_action"], a["action.x"].astype(np.int64), axis=-1)
    assert (taken == 1).all()
