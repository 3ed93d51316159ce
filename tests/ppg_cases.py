"""The Phasic-Policy-Gradient cases of tests/golden/ppg.npz (gen_golden.py PPG_CASES / ppg_entry_arrays, restated: the generator
imports the reference and stays in the build container)."""
import numpy as np

from srl_amd.runtime import synthetic

PPG_CASES = {
    "aux": (dict(obs_dim=4, action_dim=[3, 2], hidden_dim=32, num_dense_layers=1, num_rnn_layers=0, popart=False, layernorm=True,
                 chunk_len=8, seed=81),
            dict(popart=False, ppg_epochs=3, max_grad_norm=5.0, beta_clone=1.0, aux_value_head_weight=0.5,
                 ppg_optimizer_config=dict(lr=1e-3)),
            dict(T=16, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1)),
    "auxmask": (dict(obs_dim=5, action_dim=4, hidden_dim=16, num_dense_layers=2, num_rnn_layers=0,
                     popart=False, layernorm=False, chunk_len=4, seed=82, activation="tanh"),
                dict(popart=False, ppg_epochs=2, beta_clone=2.0, aux_value_head_weight=1.0, ppg_optimizer_config=dict(lr=5e-4)),
                dict(T=8, B=5, obs_spec={"obs": ((5,), "f32")}, action_dims=4, p_done=0.15, available_action=True)),
    "auxpa": (dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=2, num_rnn_layers=0, popart=True, layernorm=True,
                   chunk_len=8, seed=83, value_dim=2),
              dict(popart=True, ppg_epochs=2, max_grad_norm=1.0, beta_clone=1.0, aux_value_head_weight=1.0),
              dict(T=16, B=4, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.1, value_dim=2)),
    "auxgru": (dict(obs_dim=4, action_dim=3, hidden_dim=16, num_dense_layers=1, num_rnn_layers=1, popart=False, layernorm=True,
                    chunk_len=4, seed=84),
               dict(popart=False, ppg_epochs=2, max_grad_norm=10.0, beta_clone=1.0, aux_value_head_weight=1.0,
                    ppg_optimizer_config=dict(lr=1e-3)),
               dict(T=8, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=3, p_done=0.1,
                    policy_state={"actor_hx": (1, 16), "critic_hx": (1, 16)})),
}


def entry_arrays(arrays, T, g, tag):
    """The cache entry of a case: the sample's first T rows of obs / policy_state / on_reset, and the value targets and info_mask the
    generator drew (stored in the fixture)."""
    e = {k: v[:T] for k, v in arrays.items() if k.startswith("obs.") or k.startswith("policy_state.") or k == "on_reset"}
    e["value"], e["info_mask"] = g[f"{tag}_entry_value"], g[f"{tag}_entry_info_mask"]
    return e


def params_of(g, tag, which):
    pre = f"{tag}_{which}_param:"
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}
