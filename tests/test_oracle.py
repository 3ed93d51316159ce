"""The oracle is pinned here: against the hand-computed vectors of the reference's own tests and against the
golden vectors generated from the real reference (tests/golden/gen_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import gae as ogae
from oracle import ppo as oppo
from oracle.net import OracleActorCritic
from oracle.trainer import OracleMappo
from srl_amd.runtime import synthetic


def test_gae_hand_vector_of_reference_test():
    """legacy/tests/modules_test.py:119-138 (gamma = lambda = 0.1, done + truncated episodes)."""
    on_reset = np.array([0, 0, 0, 1, 0, 0, 1, 0, 0], dtype=np.float32)
    rew = np.array([1, 2, 0, 1, 3, 0, 1, 2, 3], dtype=np.float32)
    value = np.array([2, 0, 1, 2, 2, 0, 1, 1, 1], dtype=np.float32)
    truncated = np.array([0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float32)
    done = np.array([0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=np.float32)
    adv = ogae.gae_trace(rew[:-1], value, truncated, done, on_reset, 0.1, 0.1)
    expect = np.array([2.1 * 0.01 - 1, 2.1, 0, -0.8 + 0.01, 1, 0, 0.111, 1.1])
    keep = 1 - on_reset[1:]
    np.testing.assert_array_almost_equal(adv * keep, expect * keep)


def test_gae_vs_tianshou_style_loop():
    """legacy/tests/modules_test.py:91-117: random [101, 8, 1], done only, atol 1e-5."""
    rng = np.random.default_rng(0)
    value = rng.standard_normal((101, 8, 1))
    rew = rng.standard_normal((100, 8, 1))
    done = rng.integers(0, 2, (101, 8, 1)).astype(np.float64)
    on_reset = np.concatenate([np.zeros_like(done[:1]), done[:-1]], axis=0)
    rew = rew * (1 - on_reset[1:])
    value = value * (1 - done)
    delta = rew + value[1:] * 0.99 * (1 - done[:-1]) - value[:-1]
    m = (1.0 - done[:-1]) * (0.99 * 0.97)
    gae, ref = 0.0, np.zeros_like(rew)
    for i in range(99, -1, -1):
        gae = delta[i] + m[i] * gae
        ref[i] = gae
    adv = ogae.gae_trace(rew, value, np.zeros_like(done), done, on_reset, 0.99, 0.97)
    assert np.abs(adv - ref).max() < 1e-5


def test_traj_gae_hand_vectors(golden):
    """legacy/tests/modules_test.py:140-178 + the reference's outputs."""
    g = golden("host.npz")
    for tag, expect in (("trunc", [2.1 * 0.01 - 1, 2.1]), ("done", [-0.8 + 0.01, 1])):
        rew, value, done, trunc = g[f"trajgae_{tag}_in"]
        adv, ret = ogae.traj_gae(list(rew[:-1]), list(value[:-1]), trunc[-1], value[-1], 0.1, 0.1)
        np.testing.assert_allclose(adv, expect, rtol=1e-6)
        np.testing.assert_allclose(adv, g[f"trajgae_{tag}_adv"], rtol=1e-6)
        np.testing.assert_allclose(ret, g[f"trajgae_{tag}_ret"], rtol=1e-6)


def test_gae_tensor_gamma_lambda_golden(golden):
    """Per-step discount / lambda tensors (reference gae.py:51-60), bit-equal to the reference's own outputs."""
    g = golden("gae_tensor.npz")
    for name in g["cases"]:
        a = {k: g[f"{name}_{k}"] for k in ("reward", "value", "done", "truncated", "on_reset", "gamma", "lambda", "ratio")}
        vm = (a["value"] * (1 - a["done"])).astype(np.float32)
        for tag, gam, lam in (("gl", a["gamma"], a["lambda"]), ("g", a["gamma"], 0.95), ("l", 0.99, a["lambda"])):
            adv = ogae.gae_trace(a["reward"][:-1], vm, a["truncated"], a["done"], a["on_reset"], gam, lam)
            assert np.array_equal(adv, g[f"{name}_{tag}_adv"]), (name, tag)
        adv = ogae.gae_trace(a["reward"][:-1], vm, a["truncated"], a["done"], a["on_reset"], a["gamma"], a["lambda"],
                             vtrace=True, imp_ratio=a["ratio"])
        assert np.array_equal(adv, g[f"{name}_vtrace_adv"]), name


def test_gae_golden(golden):
    g = golden("gae.npz")
    for name in g["cases"]:
        a = {k: g[f"{name}_{k}"] for k in ("reward", "value", "done", "truncated", "on_reset")}
        for tag, gam, lam in (("a", 0.99, 0.97), ("b", 0.9, 0.5)):
            adv, ret = ogae.adv_and_value_target(a["reward"], a["value"], a["truncated"], a["done"], a["on_reset"], gam,
                                                 lam)
            assert np.array_equal(adv, g[f"{name}_{tag}_adv"]) and np.array_equal(ret, g[f"{name}_{tag}_ret"])
        v = (a["value"] * (1 - a["done"])).astype(np.float32)
        adv = ogae.gae_trace(a["reward"][:-1], v, a["truncated"], a["done"], a["on_reset"], 0.99, 0.97, vtrace=True,
                             imp_ratio=g[f"{name}_ratio"])
        assert np.array_equal(adv, g[f"{name}_vtrace_adv"])


def test_masked_normalization_golden_and_nanmean(golden):
    g = golden("norm.npz")
    x, mask = g["x"], g["mask"]
    for tag, m, unb in (("masked", mask, False), ("nomask", None, False), ("unbiased", mask, True)):
        assert np.array_equal(oppo.masked_normalization(x, m, unbiased=unb), g[f"{tag}_out"])
    # legacy/tests/modules_test.py:36-46,267-271: vs nanmean / nanstd to 6 decimals on the unmasked entries
    xc = x.astype(np.float64).copy()
    xc[mask == 0] = np.nan
    ref = (x - np.nanmean(xc)) / (np.nanstd(xc) + 1e-5)
    np.testing.assert_almost_equal(oppo.masked_normalization(x, mask) * mask, ref * mask, decimal=6)
    # all-reduced statistics override (data-parallel path)
    n, s, q = oppo.masked_stats(x, mask)
    half = x.shape[0] // 2
    parts = [oppo.masked_stats(x[:half], mask[:half]), oppo.masked_stats(x[half:], mask[half:])]
    tot = tuple(sum(p[i] for p in parts) for i in range(3))
    assert np.allclose(tot, (n, s, q), rtol=1e-13)
    assert np.array_equal(oppo.masked_normalization(x[:half], mask[:half], stats=(n, s, q)),
                          oppo.masked_normalization(x, mask)[:half])


def test_ppo_loss_golden(golden):
    g = golden("loss.npz")
    names = list(g["stat_names"])
    t = lambda k, grad=False: torch.from_numpy(g[k]).clone().requires_grad_(grad)
    for combo in g["combos"]:
        vl, cv, dc = combo.split("_")
        nlp, v, ent = t("new_lp", True), t("value", True), t("entropy", True)
        loss, stats = oppo.ppo_loss(nlp, t("old_lp"), v, t("old_value"), t("adv"), t("ret"), ent, t("mask"),
                                    dual_clip=dc == "1", value_loss=vl, clip_value=cv == "1",
                                    value_loss_config=dict(delta=10.0) if vl == "huber" else {})
        loss.backward()
        stats["loss"] = loss.item()
        ref = dict(zip(names, g[f"{combo}_stats"]))
        for k, val in ref.items():
            assert abs(stats[k] - val) <= 1e-6 * max(1, abs(val)), (combo, k)
        assert torch.allclose(nlp.grad, torch.from_numpy(g[f"{combo}_d_new_lp"]), rtol=1e-6, atol=1e-9)
        assert torch.allclose(v.grad, torch.from_numpy(g[f"{combo}_d_value"]), rtol=1e-6, atol=1e-9)
        assert torch.allclose(ent.grad, torch.from_numpy(g[f"{combo}_d_entropy"]), rtol=1e-6, atol=1e-9)


def test_full_step_golden_c1(golden):
    """OracleMappo reproduces the reference trainer's statistics and post-step parameters (config #1 shapes)."""
    g = golden("steps_mlp.npz")
    pargs = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False,
                 layernorm=False, shared_backbone=False, chunk_len=8)
    net = OracleActorCritic(**pargs)
    net.load_state_dict({k[len("c1_init_param:"):]: g[k] for k in g.files if k.startswith("c1_init_param:")})
    tr = OracleMappo(net, popart=False, optimizer_config=dict(lr=3e-4))
    names = list(g["c1_stat_names"])
    for step in range(3):
        arrays = synthetic.make_sample_arrays(seed=100 + step, T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2,
                                              p_done=0.05)
        stats, out = tr.step(arrays)
        ref = dict(zip(names, g[f"c1_step{step}_stats"]))
        for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "clip_ratio"):
            assert abs(stats[k] - ref[k]) <= 2e-5 * max(1.0, abs(ref[k])), (step, k)
        if step == 0:
            assert np.array_equal(out["ret"], g["c1_step0_ret"])
    sd = net.state_dict()
    for k in sd:
        assert np.allclose(sd[k].numpy(), g[f"c1_step2_param:{k}"], rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("tag,pargs,targs,n_steps,p_trunc", [
    ("pa", dict(popart=True), dict(popart=True, optimizer_config=dict(lr=3e-4)), 3, None),
    ("pa2", dict(popart=True, layernorm=True, shared_backbone=True),
     dict(popart=True, clip_value=True, dual_clip=False, value_loss='huber', value_loss_config=dict(delta=10.0),
          value_loss_weight=1.0, ppo_epochs=2, optimizer_config=dict(lr=5e-4), max_grad_norm=40.0), 2, None),
    ("vt", dict(), dict(popart=False, vtrace=True, optimizer_config=dict(lr=3e-4)), 2, 0.0),
    ("vtpa", dict(popart=True), dict(popart=True, vtrace=True, max_grad_norm=10.0, optimizer_config=dict(lr=1e-3)), 2, 0.0),
])
def test_full_step_golden_popart_vtrace(golden, tag, pargs, targs, n_steps, p_trunc):
    """PopArt head (float64 running statistics, normalised targets) and V-trace: OracleMappo reproduces the
    reference trainer's statistics, value targets and post-step state (fixtures: gen_golden.py gen_popart)."""
    g = golden("steps_popart.npz")
    base = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False,
                layernorm=False, shared_backbone=False, chunk_len=8)
    net = OracleActorCritic(**dict(base, **pargs))
    pre = f"{tag}_init_param:"
    net.load_state_dict({k[len(pre):]: g[k] for k in g.files if k.startswith(pre)})
    tr = OracleMappo(net, **targs)
    names = list(g[f"{tag}_stat_names"])
    for step in range(n_steps):
        arrays = synthetic.make_sample_arrays(seed=100 + step, T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2,
                                              p_done=0.05, p_trunc=p_trunc)
        stats, out = tr.step(arrays)
        ref = dict(zip(names, g[f"{tag}_step{step}_stats"]))
        keys = ["policy_loss", "value_loss", "entropy", "grad_norm", "clip_ratio", "value_targets"]
        keys += ["denorm_value"] if targs["popart"] else []
        for k in keys:
            assert abs(stats[k] - ref[k]) <= 2e-5 * max(1.0, abs(ref[k])), (tag, step, k)
        if step == 0:
            assert np.allclose(out["ret"], g[f"{tag}_step0_ret"], rtol=1e-5, atol=1e-6)
    sd = net.state_dict()
    for k in sd:
        assert np.allclose(sd[k].numpy(), g[f"{tag}_step{n_steps - 1}_param:{k}"], rtol=1e-4, atol=1e-6), k


def test_full_step_golden_recurrent(golden):
    """GRU backbone with auto reset and chunked analysis: OracleMappo against the reference (gen_golden.py gen_rnn)."""
    g = golden("steps_rnn.npz")
    pargs = dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=1, popart=False,
                 layernorm=True, shared_backbone=True, chunk_len=8)
    net = OracleActorCritic(**pargs)
    net.load_state_dict({k[len("gru_init_param:"):]: g[k] for k in g.files if k.startswith("gru_init_param:")})
    tr = OracleMappo(net, popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0)
    names = list(g["gru_stat_names"])
    for step in range(3):
        arrays = synthetic.make_sample_arrays(seed=100 + step, T=32, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2,
                                              p_done=0.08, policy_state={"hx": (1, 32)})
        stats, _ = tr.step(arrays)
        ref = dict(zip(names, g[f"gru_step{step}_stats"]))
        for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "clip_ratio"):
            assert abs(stats[k] - ref[k]) <= 2e-5 * max(1.0, abs(ref[k])), (step, k)
    sd = net.state_dict()
    for k in sd:
        assert np.allclose(sd[k].numpy(), g[f"gru_step2_param:{k}"], rtol=1e-4, atol=1e-6), k


def test_oracle_data_parallel_emulation_reduces_to_single_rank():
    """OracleMappo.step_dp (the reference's DDP semantics on the CPU) with ONE rank is the plain step; with two ranks
    whose columns are copies of each other the gradient mean and the summed statistics change nothing either."""
    pargs = dict(obs_dim=4, action_dim=3, hidden_dim=16, num_dense_layers=1, num_rnn_layers=0, popart=True, layernorm=True,
                 shared_backbone=False, chunk_len=8)
    targs = dict(popart=True, ppo_epochs=2, clip_value=True, dual_clip=False, value_loss="huber",
                 value_loss_config=dict(delta=10.0), optimizer_config=dict(lr=1e-3), max_grad_norm=5.0)
    from srl_amd.algorithm.netspec import build_netspec
    _, init = build_netspec(**{k: v for k, v in pargs.items() if k != "chunk_len"}, seed=3)
    arrays = synthetic.make_sample_arrays(seed=9, T=16, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=3, p_done=0.1)
    nets, trainers = [], []
    for _ in range(3):
        net = OracleActorCritic(**pargs)
        net.load_state_dict({k: v.numpy() for k, v in init.items()})
        nets.append(net)
        trainers.append(OracleMappo(net, **targs))
    s_plain, _ = trainers[0].step(arrays)
    s_one, _ = trainers[1].step_dp([arrays])
    s_two, _ = trainers[2].step_dp([arrays, {k: v.copy() for k, v in arrays.items()}])
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "denorm_value"):
        assert abs(s_plain[k] - s_one[0][k]) <= 1e-6 * max(1.0, abs(s_plain[k])), k
        assert abs(s_plain[k] - s_two[1][k]) <= 1e-5 * max(1.0, abs(s_plain[k])), k
    for k, v in nets[0].state_dict().items():
        assert np.allclose(v.numpy(), nets[1].state_dict()[k].numpy(), rtol=1e-6, atol=1e-7), k
        if "_RunningMeanStd__" not in k:  # doubled data moves the PopArt EMA identically (means), so everything matches
            assert np.allclose(v.numpy(), nets[2].state_dict()[k].numpy(), rtol=1e-4, atol=1e-6), k


# ------------------------------------------------------------------------------------------------ PPG
@pytest.mark.parametrize("tag", ["aux", "auxmask", "auxpa", "auxgru"])
def test_ppg_auxiliary_phase_oracle_vs_reference_golden(tag, golden):
    """oracle/ppg.py + OracleActorCritic.analyze_aux against what the reference's runnable PPG pieces produced
    (tests/golden/gen_golden.py gen_ppg): both analysis targets of the `actor-critic-auxiliary` policy, then the auxiliary phase
    -- loss terms and gradient norm of every epoch, every parameter after the epochs."""
    import torch
    from oracle.net import OracleActorCritic
    from oracle.ppg import OraclePPGAux
    from ppg_cases import PPG_CASES, entry_arrays, params_of
    from srl_amd.runtime import synthetic
    g = golden("ppg.npz")
    pargs, targs, skw = PPG_CASES[tag]
    T = skw["T"]
    arrays = synthetic.make_sample_arrays(seed=300, **skw)
    net = OracleActorCritic(**pargs)
    net.load_state_dict(params_of(g, tag, "init"))
    f32 = lambda a: torch.from_numpy(np.asarray(a)).float()
    obs = {k[4:]: f32(v[:T]) for k, v in arrays.items() if k.startswith("obs.")}
    ps = [f32(arrays[n][:T]) for n in ("policy_state.actor_hx", "policy_state.critic_hx") if n in arrays] or None
    with torch.no_grad():
        lp, value, ent, _ = net.analyze(obs, f32(arrays["action.x"][:T]), f32(arrays["on_reset"][:T]), ps)
    unchunk_aux = net._aux_value
    if net.num_rnn_layers:  # analyze() returned trajectory-major values; the auxiliary values are still chunk-major
        n = T // net.chunk_len
        unchunk_aux = torch.cat(torch.split(unchunk_aux, unchunk_aux.shape[1] // n, dim=1), dim=0)
    for name, got in (("new_lp", lp), ("value", value), ("entropy", ent), ("aux", unchunk_aux)):
        ref = g[f"{tag}_p1_{name}"]
        assert np.abs(got.detach().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), name
    e = entry_arrays(arrays, T, g, tag)
    aux = OraclePPGAux(net, beta_clone=targs.get("beta_clone", 1), aux_value_head_weight=targs.get("aux_value_head_weight", 1),
                       max_grad_norm=targs.get("max_grad_norm"), popart=targs.get("popart", False),
                       ppg_optimizer_config=targs.get("ppg_optimizer_config", {}))
    aux.enter(e)
    for h, d in enumerate(aux.old):
        assert np.abs(d.numpy() - g[f"{tag}_p2_logq{h}"]).max() <= 1e-5
    for k, v in params_of(g, tag, "pert").items():
        net.params[k].data.copy_(torch.from_numpy(v).to(net.params[k].dtype))
    names = list(g["term_names"])
    for ep in range(targs["ppg_epochs"]):
        o = aux.epoch(e)
        ref = dict(zip(names, g[f"{tag}_epoch{ep}_terms"]))
        for k in ("auxiliary_value_loss", "value_head_loss", "policy_distance", "loss"):
            assert abs(o[k] - ref[k]) <= 1e-5 * max(1.0, abs(ref[k])), (ep, k, o[k], ref[k])
        if ref["grad_norm"] >= 0:
            assert abs(o["grad_norm"] - ref["grad_norm"]) <= 2e-5 * max(1.0, ref["grad_norm"]), ep
    for k, v in params_of(g, tag, "final").items():
        assert np.abs(net.params[k].detach().numpy() - v).max() <= 2e-5, k
