"""Phasic Policy Gradient on the device (second half): the `actor-critic-auxiliary` policy's two analysis targets and the
`mappg` trainer's auxiliary phase against the golden vectors the reference's runnable pieces produced (tests/golden/gen_golden.py
gen_ppg; tests/test_oracle.py checks the CPU oracle against the same vectors), the fused auxiliary-loss kernel against the oracle
on ragged sizes, and the full `mappg.step` flow (phase-1 glue: parity unpinned, see mappg.py) for what it can be held to."""
import numpy as np
import pytest
import torch

import srl_amd
from ppg_cases import PPG_CASES, entry_arrays, params_of
from srl_amd import hip
from srl_amd.api import config, policy as policy_api, trainer as trainer_api
from srl_amd.namedarray import NamedArray
from srl_amd.runtime import synthetic

srl_amd.register_all()
pytestmark = pytest.mark.gpu


def _close(a, b, tol, scale=1.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool((np.abs(a - b) <= tol * np.maximum(np.abs(b), scale)).all())


def _make(tag):
    pargs, targs, skw = PPG_CASES[tag]
    trainer = trainer_api.make(config.Trainer("mappg", args=targs), config.Policy("actor-critic-auxiliary", args=pargs))
    return trainer, pargs, targs, skw


def _load(trainer, params):
    trainer.policy.load_checkpoint(dict(steps=trainer.policy.version, state_dict={k: torch.from_numpy(np.asarray(v)) for k, v in params.items()}))


@pytest.mark.parametrize("tag", list(PPG_CASES))
def test_ppg_analysis_targets_and_auxiliary_phase_match_reference_golden(tag, golden):
    from srl_amd.algorithm.mappg import _CacheEntry
    g = golden("ppg.npz")
    trainer, pargs, targs, skw = _make(tag)
    pol = trainer.policy
    init = params_of(g, tag, "init")
    assert list(pol.get_checkpoint()["state_dict"]) == list(init)   # the reference's state_dict keys, in its order
    _load(trainer, init)
    T = skw["T"]
    arrays = synthetic.make_sample_arrays(seed=300, **skw)
    sample = synthetic.to_sample_batch(arrays)
    # ---- analyze(target="ppg_ppo_phase"): actor_critic_policy.py:392-415
    r1 = pol.analyze(sample[:T], target="ppg_ppo_phase")
    assert _close(r1.new_action_log_probs.cpu().numpy(), g[f"{tag}_p1_new_lp"], 1e-5)
    assert _close(r1.state_values.cpu().numpy(), g[f"{tag}_p1_value"], 1e-5)
    assert _close(r1.entropy.cpu().numpy(), g[f"{tag}_p1_entropy"], 1e-5)
    assert _close(r1.aux_values.cpu().numpy(), g[f"{tag}_p1_aux"], 1e-5)
    # ---- analyze(target="ppg_aux_phase") on the cache entry: :417-435
    e = entry_arrays(arrays, T, g, tag)
    obs = {k[4:]: v for k, v in e.items() if k.startswith("obs.")}
    ps = {k[len("policy_state."):]: v for k, v in e.items() if k.startswith("policy_state.")}
    entry_sample = NamedArray(obs=NamedArray(**obs), policy_state=NamedArray(**ps) if ps else None, on_reset=e["on_reset"],
                              info_mask=e["info_mask"])
    r2 = pol.analyze(entry_sample, target="ppg_aux_phase")
    for h, d in enumerate(r2.action_dists):
        ref = g[f"{tag}_p2_logq{h}"]
        live = ref > -1e9  # unavailable actions sit at ~ -1e10 on both sides
        assert np.abs(d.logits.cpu().numpy() - ref)[live].max() <= 1e-5 and (d.logits.cpu().numpy()[~live] < -1e9).all(), h
    assert _close(r2.auxiliary_value.cpu().numpy(), g[f"{tag}_p2_aux"], 1e-5)
    assert _close(r2.predicted_value.cpu().numpy(), g[f"{tag}_p2_pred"], 1e-5)
    # ---- the auxiliary phase: distributions kept under the initial parameters, epochs from the perturbed ones (gen_ppg)
    entry = _CacheEntry(obs, ps or None, e["info_mask"], e["on_reset"], e["value"])
    trainer.enter_aux_phase(entry)
    _load(trainer, params_of(g, tag, "pert"))
    names = list(g["term_names"])
    v0 = pol.version
    for ep in range(targs["ppg_epochs"]):
        m = trainer.aux_epoch(entry)
        ref = dict(zip(names, g[f"{tag}_epoch{ep}_terms"]))
        for k in ("auxiliary_value_loss", "value_head_loss", "policy_distance"):
            assert abs(getattr(m, k) - ref[k]) <= 1e-5 * max(abs(ref[k]), 1e-2), (ep, k, getattr(m, k), ref[k])
        if ref["grad_norm"] >= 0:
            assert abs(trainer.last_aux_grad_norm - ref["grad_norm"]) <= 2e-5 * max(ref["grad_norm"], 1e-2), (ep, trainer.last_aux_grad_norm)
    assert pol.version == v0 + targs["ppg_epochs"]   # :243 inc_version per auxiliary epoch
    sd = pol.get_checkpoint()["state_dict"]
    for k, v in params_of(g, tag, "final").items():
        assert np.abs(sd[k].numpy() - v).max() <= 2e-5, (k, np.abs(sd[k].numpy() - v).max())


@pytest.mark.parametrize("n,heads,vd,mask", [(1, [2], 1, False), (77, [3, 2], 1, True), (1000, [9], 3, True), (4099, [4, 3, 2], 2, False)])
def test_aux_loss_kernel_vs_oracle(n, heads, vd, mask):
    """srl_ppg_aux_loss_fwd_bwd and srl_categorical_log_softmax on their own: terms and all three gradients against autograd of
    oracle/ppg.py's restatement (ragged row counts, several heads, value_dim > 1, an availability mask, rows with done = 1)."""
    from oracle.ppg import aux_loss
    rng = np.random.default_rng(n)
    A = sum(heads)
    z_old = torch.from_numpy(rng.standard_normal((n, A)).astype(np.float32))
    z_new = (z_old + 0.3 * torch.from_numpy(rng.standard_normal((n, A)).astype(np.float32))).requires_grad_(True)
    avail = None
    if mask:
        av = rng.random((n, A)) < 0.7
        s = 0
        for d in heads:
            av[:, s] = True  # at least one legal action per head
            s += d
        avail = torch.from_numpy(av.astype(np.uint8))
    aux = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).requires_grad_(True)
    pred = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32)).requires_grad_(True)
    tgt = torch.from_numpy(rng.standard_normal((n, vd)).astype(np.float32))
    done = torch.from_numpy((rng.random((n, 1)) < 0.25).astype(np.uint8))
    if n > 1:
        done[0] = 0
    else:
        done[:] = 0
    beta, vhw = 1.7, 0.6
    mz = lambda z: z if avail is None else z.masked_fill(avail == 0, -1e10)
    split = lambda z: [torch.log_softmax(p, -1) for p in torch.split(mz(z), heads, dim=-1)]
    with torch.no_grad():
        old = split(z_old)
    loss, terms = aux_loss([o[None] for o in old], [q[None] for q in split(z_new)], aux[None], pred[None], tgt[None], done[None].float(),
                           beta, vhw)
    loss.backward()
    dev = "cuda:0"
    logq = torch.empty((n, A), device=dev)
    hip.categorical_log_softmax(z_old.to(dev), None if avail is None else avail.to(dev), heads, logq)
    live = torch.cat(old, -1) > -1e9
    assert float((logq.cpu() - torch.cat(old, -1))[live].abs().max()) <= 1e-5
    count = torch.zeros(3, dtype=torch.float64, device=dev)
    hip.masked_stats(torch.zeros(n, device=dev), done.reshape(n).to(dev), count, mask_invert=True)
    d_logits, d_aux, d_pred = torch.empty((n, A), device=dev), torch.empty((n, vd), device=dev), torch.empty((n, vd), device=dev)
    out = torch.empty(3, dtype=torch.float64, device=dev)
    hip.ppg_aux_loss_fwd_bwd(logq, z_new.detach().to(dev), None if avail is None else avail.to(dev), heads, aux.detach().to(dev),
                             pred.detach().to(dev), tgt.to(dev), done.reshape(n).to(dev), count[0:1], beta, vhw, d_logits, d_aux, d_pred, out)
    got = out.cpu().numpy()
    for i, k in enumerate(("auxiliary_value_loss", "value_head_loss", "policy_distance")):
        assert abs(got[i] - float(terms[k].detach())) <= 1e-5 * max(abs(float(terms[k].detach())), 1e-3), (k, got[i], float(terms[k].detach()))
    for name, g_dev, g_ref in (("logits", d_logits, z_new.grad), ("aux", d_aux, aux.grad), ("pred", d_pred, pred.grad)):
        scale = float(g_ref.abs().max())
        assert float((g_dev.cpu() - g_ref).abs().max()) <= 1e-5 * max(scale, 1e-8), (name, float((g_dev.cpu() - g_ref).abs().max()), scale)


def test_mappg_step_flow_cache_and_checkpoint():
    """`mappg.step` end to end: phase 1 IS the MAPPO step (same statistics under `ppo_` names and the same parameters as a `mappo`
    trainer fed the same samples, the auxiliary head untouched), the local cache fills to `ppo_iterations`, the auxiliary phase then
    runs `ppg_epochs` epochs per cached sample (policy version, `ppg_` statistics, parameters move, cache cleared); the auxiliary
    optimiser's state travels in the checkpoint (phasic_policy_gradient.py:120-128)."""
    pargs = dict(obs_dim=4, action_dim=[3, 2], hidden_dim=32, num_dense_layers=1, num_rnn_layers=0, popart=True, layernorm=True,
                 chunk_len=8, seed=5)
    targs = dict(popart=True, ppo_epochs=2, ppo_iterations=2, ppg_epochs=2, max_grad_norm=5.0, optimizer_config=dict(lr=1e-3),
                 ppg_optimizer_config=dict(lr=5e-4), aux_value_head_weight=0.5)
    skw = dict(T=16, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1)
    ppg = trainer_api.make(config.Trainer("mappg", args=targs), config.Policy("actor-critic-auxiliary", args=pargs))
    ppo = trainer_api.make(config.Trainer("mappo", args=dict(popart=True, ppo_epochs=2, max_grad_norm=5.0, optimizer_config=dict(lr=1e-3))),
                           config.Policy("actor-critic-auxiliary", args=pargs))
    s0 = synthetic.to_sample_batch(synthetic.make_sample_arrays(seed=1, **skw))
    v0 = ppg.policy.version
    r_ppg = ppg.step(s0)
    r_ppo = ppo.step(synthetic.to_sample_batch(synthetic.make_sample_arrays(seed=1, **skw)))
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        assert abs(r_ppg.stats[f"ppo_{k}"] - r_ppo.stats[k]) <= 1e-6 * max(abs(r_ppo.stats[k]), 1e-2), k
    a, b = ppg.policy.get_checkpoint()["state_dict"], ppo.policy.get_checkpoint()["state_dict"]
    # (the fused chains sum their parameter gradients with float atomics: equal to summation order, not bitwise -- and Adam's first
    # steps move a parameter by ~lr * sign(gradient), so an element whose gradient is zero to rounding may differ by 2 lr per
    # epoch between two runs: all but a handful of elements agree to 1e-6, none differs by more than the steps taken)
    far = total = 0
    for k in a:
        d = (a[k].double() - b[k].double()).abs()
        far += int((d > 1e-6).sum())
        total += d.numel()
        assert float(d.max()) <= 2 * 2 * 1e-3 * 1.01, (k, float(d.max()))
    assert far <= max(2, total // 200), (far, total)
    assert len(ppg._cache) == 1 and "ppg_policy_distance" not in r_ppg.stats and ppg.policy.version == v0 + 1
    aux_head0 = a["auxiliary_value_head.weight"].clone()
    r2 = ppg.step(synthetic.to_sample_batch(synthetic.make_sample_arrays(seed=2, **skw)))
    assert len(ppg._cache) == 0 and ppg._aux_steps == 4   # 2 cached samples x 2 epochs
    assert ppg.policy.version == v0 + 2 + 4               # one per PPO step (:184... mappo.py:305-307 here), one per auxiliary epoch
    for k in ("ppg_auxiliary_value_loss", "ppg_value_head_loss", "ppg_policy_distance"):
        assert np.isfinite(r2.stats[k]) and r2.stats[k] >= 0, k
    c = ppg.policy.get_checkpoint()["state_dict"]
    assert not torch.equal(c["auxiliary_value_head.weight"], aux_head0)   # only the auxiliary phase trains that head
    ck = ppg.get_checkpoint()
    st = ck["aux_optimizer_state_dict"]
    # torch's Adam layout (phasic_policy_gradient.py:120-128 stores `ppg_aux_optimizer.state_dict()`), parameters in the reference's order
    names = ppg.policy.net.ref_names()
    ia = names.index("auxiliary_value_head.weight")
    assert set(st) == {"state", "param_groups"} and float(st["state"][ia]["step"]) == 4
    assert float(st["state"][ia]["exp_avg_sq"].abs().sum()) > 0 and st["param_groups"][0]["lr"] == 5e-4
    fresh = trainer_api.make(config.Trainer("mappg", args=targs), config.Policy("actor-critic-auxiliary", args=pargs))
    fresh.load_checkpoint(ck)
    assert fresh._aux_steps == 4 and torch.equal(fresh._aux_m, ppg._aux_m) and torch.equal(fresh._aux_v, ppg._aux_v)
    # ... and a state dict GENERATED BY torch.optim.Adam over parameters of the reference's shapes loads (what a reference PPG
    # checkpoint carries), and ours loads into torch's
    sd = ppg.policy.get_checkpoint()["state_dict"]
    tparams = [torch.nn.Parameter(sd[n].clone().float()) for n in names] + [torch.nn.Parameter(torch.zeros(1), requires_grad=False)
                                                                              for _ in range(3)]
    topt = torch.optim.Adam(tparams, lr=2.5e-4, betas=(0.8, 0.95), eps=1e-6)
    gen = torch.Generator().manual_seed(3)
    for _ in range(3):
        for p_ in tparams[:len(names)]:
            p_.grad = torch.randn(p_.shape, generator=gen)
        topt.step()
    ck2 = dict(ck, aux_optimizer_state_dict=topt.state_dict())
    fresh.load_checkpoint(ck2)
    assert fresh._aux_steps == 3 and fresh._aux_lr == 2.5e-4 and fresh._aux_betas == (0.8, 0.95) and fresh._aux_eps == 1e-6
    m_ref = fresh.policy.net.reference_to_flat({n: topt.state[p_]["exp_avg"] for n, p_ in zip(names, tparams)})
    assert torch.equal(fresh._aux_m.cpu(), m_ref.cpu())
    torch.optim.Adam(tparams, lr=1e-3).load_state_dict(ck["aux_optimizer_state_dict"])   # torch accepts ours
    # episode-info averages keep their names, only the PPO step's statistics get the `ppo_` prefix (:242-253)
    assert "ppo_policy_loss" in r2.stats and "ppo_grad_norm" in r2.stats and "frames" in r2.stats
    assert not any(k.startswith("ppo_episode") for k in r2.stats)
    with pytest.raises(ValueError):
        trainer_api.make(config.Trainer("mappg", args=dict(targs, recompute_adv_among_epochs=True)),
                         config.Policy("actor-critic-auxiliary", args=pargs))
