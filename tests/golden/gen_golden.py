"""Generate the golden fixtures under tests/golden/ by running the REAL reference (read-only at
/root/reference) in this build container.  The reference cannot travel to the GPU box, so the
vectors are committed as small .npz files together with this script.

Run (from the repo root):
    CUDA_VISIBLE_DEVICES="" PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:. python3 tests/golden/gen_golden.py

Harness-side shims only: numpy>=2 alias, MagicMock for absent import-time deps, and a CPU
pass-through in place of the CUDA prefetcher (api/trainer.py:199-228 needs a GPU to construct).
While generating, the script also asserts that this repo's oracle/ and initialiser agree with the
reference on the same inputs; fixtures are only written if they do.
"""
import hashlib
import os
import sys

import numpy as np

np.bool8 = np.bool_
from unittest import mock

for m in ["gym", "gym.spaces", "redis", "redis.backoff", "redis.retry", "wandb", "zmq", "blosc"]:
    sys.modules[m] = mock.MagicMock()
sys.modules.setdefault("mock", mock)
import torch

torch.set_num_threads(1)  # deterministic CPU reductions
import api.config
import api.policy
import api.trainer
from api.env_utils import DiscreteAction
from base.namedarray import NamedArray, recursive_apply
import base.namedarray as ref_namedarray


class CPUPrefetcher:

    def push(self, sample):
        return sample, recursive_apply(sample, lambda x: torch.from_numpy(x).float())


api.trainer.PyTorchGPUPrefetcher = CPUPrefetcher
import legacy.algorithm.ppo.mappo as mappo

mappo.PyTorchGPUPrefetcher = CPUPrefetcher
import legacy.algorithm.modules as modules
from legacy.algorithm.ppo.actor_critic_policies.actor_critic_policy import PPORolloutAnalyzedResult
from legacy.algorithm.ppo.mappo import SampleAnalyzedResult

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gae as ogae
from oracle import ppo as oppo
from oracle.net import OracleActorCritic
from oracle.trainer import OracleMappo
from srl_amd.algorithm.netspec import build_netspec
from srl_amd.runtime import synthetic

TORCH_VERSION = torch.__version__


def save(name, **arrays):
    arrays["torch_version"] = np.array(TORCH_VERSION)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


def state_sha(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().cpu().numpy()).tobytes())
    return h.hexdigest()


def ref_sample(arrays):
    """flat dict -> the reference's SampleBatch."""
    obs = NamedArray(**{k[4:]: v for k, v in arrays.items() if k.startswith("obs.")})
    ps = {k[len("policy_state."):]: v for k, v in arrays.items() if k.startswith("policy_state.")}
    return api.trainer.SampleBatch(obs=obs, policy_state=NamedArray(**ps) if ps else None,
                                   on_reset=arrays["on_reset"], done=arrays["done"],
                                   truncated=arrays["truncated"], action=DiscreteAction(arrays["action.x"]),
                                   reward=arrays["reward"],
                                   analyzed_result=PPORolloutAnalyzedResult(
                                       log_probs=arrays["analyzed_result.log_probs"],
                                       value=arrays["analyzed_result.value"]),
                                   policy_version_steps=arrays["policy_version_steps"],
                                   info_mask=arrays["info_mask"])


# ------------------------------------------------------------------------------------------------ G1/G2
def gen_gae():
    out = {}
    # the hand-computed case of legacy/tests/modules_test.py:119-138 (inputs restated; outputs from the reference)
    on_reset = np.array([0, 0, 0, 1, 0, 0, 1, 0, 0], dtype=np.float32)
    rew = np.array([1, 2, 0, 1, 3, 0, 1, 2, 3], dtype=np.float32)
    value = np.array([2, 0, 1, 2, 2, 0, 1, 1, 1], dtype=np.float32)
    truncated = np.array([0, 0, 1, 0, 0, 0, 0, 0, 0], dtype=np.float32)
    done = np.array([0, 0, 0, 0, 0, 1, 0, 0, 0], dtype=np.float32)
    t = torch.from_numpy
    adv = modules.gae_trace(reward=t(rew)[:-1], value=t(value), truncated=t(truncated), done=t(done),
                            on_reset=t(on_reset), gamma=0.1, lmbda=0.1).numpy()
    out.update(hand_on_reset=on_reset, hand_reward=rew, hand_value=value, hand_truncated=truncated, hand_done=done,
               hand_adv=adv)

    cases = [("small", 32, 8, 1, 0.05, 21), ("mid", 128, 64, 1, 0.02, 22), ("nc3", 32, 8, 3, 0.1, 23),
             ("dense_done", 16, 5, 1, 0.4, 24)]
    names = []
    for name, T, B, Nc, p, seed in cases:
        arr = synthetic.make_sample_arrays(seed=seed, T=T, B=B, obs_spec={}, action_dims=2, p_done=p,
                                           value_dim=Nc)
        f = lambda k: torch.from_numpy(arr[k]).float()
        # reference path: _compute_adv_and_value_target without popart (mappo.py:118-144)
        v_masked = f("analyzed_result.value") * (1 - f("done"))
        for tag, g, l in [("a", 0.99, 0.97), ("b", 0.9, 0.5)]:
            adv = modules.gae_trace(f("reward")[:-1], v_masked, f("truncated"), f("done"), f("on_reset"), gamma=g,
                                    lmbda=l)
            ret = adv + v_masked[:-1]
            o_adv, o_ret = ogae.adv_and_value_target(arr["reward"], arr["analyzed_result.value"], arr["truncated"],
                                                     arr["done"], arr["on_reset"], g, l)
            assert np.array_equal(o_adv, adv.numpy()) and np.array_equal(o_ret, ret.numpy()), name
            out[f"{name}_{tag}_adv"], out[f"{name}_{tag}_ret"] = adv.numpy(), ret.numpy()
        # v-trace variant (rho = c = 1) with a synthetic importance ratio
        rng = np.random.default_rng(7)
        ratio = np.exp(0.3 * rng.standard_normal((T, B, 1))).astype(np.float32)
        ratio[arr["truncated"][:-1] == 1] = 1.0  # keeps the reference's debug assertion gae.py:77 valid under v-trace
        adv_v = modules.gae_trace(f("reward")[:-1], v_masked, f("truncated"), f("done"), f("on_reset"), gamma=0.99,
                                  lmbda=0.97, vtrace=True, imp_ratio=torch.from_numpy(ratio), rho=1.0, c=1.0)
        o_adv_v = ogae.gae_trace(arr["reward"][:-1], v_masked.numpy(), arr["truncated"], arr["done"], arr["on_reset"],
                                 0.99, 0.97, vtrace=True, imp_ratio=ratio)
        assert np.array_equal(o_adv_v, adv_v.numpy())
        out[f"{name}_ratio"], out[f"{name}_vtrace_adv"] = ratio, adv_v.numpy()
        for k in ["reward", "analyzed_result.value", "done", "truncated", "on_reset"]:
            out[f"{name}_{k.split('.')[-1]}"] = arr[k]
        names.append(name)
    out["cases"] = np.array(names)
    save("gae.npz", **out)


def gen_gae_tensor():
    """G1, tensor-valued discount / lambda ([T, B, 1] float32 tensors, gae.py:51-60): both tensors, only gamma, only
    lambda (the other a python float), with and without V-trace, one and three value channels."""
    out, names = {}, []
    for name, T, B, Nc, p, seed in [("tsmall", 32, 8, 1, 0.05, 31), ("tmid", 129, 36, 1, 0.03, 32), ("tnc3", 24, 8, 3, 0.1, 33)]:
        arr = synthetic.make_sample_arrays(seed=seed, T=T, B=B, obs_spec={}, action_dims=2, p_done=p, value_dim=Nc)
        f = lambda k: torch.from_numpy(arr[k]).float()
        v_masked = f("analyzed_result.value") * (1 - f("done"))
        rng = np.random.default_rng(seed)
        gam = rng.uniform(0.9, 1.0, size=(T, B, 1)).astype(np.float32)
        lam = rng.uniform(0.5, 1.0, size=(T, B, 1)).astype(np.float32)
        ratio = np.exp(0.3 * rng.standard_normal((T, B, 1))).astype(np.float32)
        ratio[arr["truncated"][:-1] == 1] = 1.0
        for tag, g, l in [("gl", gam, lam), ("g", gam, 0.95), ("l", 0.99, lam)]:
            tg = torch.from_numpy(g) if isinstance(g, np.ndarray) else g
            tl = torch.from_numpy(l) if isinstance(l, np.ndarray) else l
            adv = modules.gae_trace(f("reward")[:-1], v_masked, f("truncated"), f("done"), f("on_reset"), gamma=tg, lmbda=tl)
            o_adv = ogae.gae_trace(arr["reward"][:-1], v_masked.numpy(), arr["truncated"], arr["done"], arr["on_reset"], g, l)
            assert np.array_equal(o_adv, adv.numpy()), (name, tag)
            out[f"{name}_{tag}_adv"] = adv.numpy()
        adv_v = modules.gae_trace(f("reward")[:-1], v_masked, f("truncated"), f("done"), f("on_reset"),
                                  gamma=torch.from_numpy(gam), lmbda=torch.from_numpy(lam), vtrace=True,
                                  imp_ratio=torch.from_numpy(ratio), rho=1.0, c=1.0)
        o_adv_v = ogae.gae_trace(arr["reward"][:-1], v_masked.numpy(), arr["truncated"], arr["done"], arr["on_reset"], gam, lam,
                                 vtrace=True, imp_ratio=ratio)
        assert np.array_equal(o_adv_v, adv_v.numpy()), name
        out[f"{name}_vtrace_adv"] = adv_v.numpy()
        out[f"{name}_gamma"], out[f"{name}_lambda"], out[f"{name}_ratio"] = gam, lam, ratio
        for k in ["reward", "analyzed_result.value", "done", "truncated", "on_reset"]:
            out[f"{name}_{k.split('.')[-1]}"] = arr[k]
        names.append(name)
    out["cases"] = np.array(names)
    save("gae_tensor.npz", **out)


# ------------------------------------------------------------------------------------------------ G3
def gen_norm():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((33, 9, 1)).astype(np.float32) * 3 + 0.7
    mask = (rng.random((33, 9, 1)) < 0.8).astype(np.float32)
    out = dict(x=x, mask=mask)
    for tag, m, unb in [("masked", mask, False), ("nomask", None, False), ("unbiased", mask, True)]:
        y = modules.masked_normalization(torch.from_numpy(x), None if m is None else torch.from_numpy(m),
                                         unbiased=unb).numpy()
        o = oppo.masked_normalization(x, m, unbiased=unb)
        assert np.array_equal(o, y), tag
        out[f"{tag}_out"] = y
        out[f"{tag}_stats"] = np.array(oppo.masked_stats(x, m), dtype=np.float64)
    save("norm.npz", **out)


# ------------------------------------------------------------------------------------------------ G4
def make_ref_trainer(policy_args, trainer_args, policy_type="actor-critic"):
    trainer = api.trainer.make(api.config.Trainer("mappo", args=trainer_args),
                               api.config.Policy(policy_type, args=policy_args))
    if policy_args.get("popart"):
        # harness shim: `popart_head` reads `self.net.module` (actor_critic_policy.py:265), which only exists once
        # DistributedDataParallel wraps the net; single-process, give the bare net a plain `module` attribute
        # (object.__setattr__: not registered as a sub-module, so state_dict / parameters are unchanged)
        object.__setattr__(trainer.policy.net, "module", trainer.policy.net)
    return trainer


C1_POLICY = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False,
                 layernorm=False, shared_backbone=False, chunk_len=8, seed=1)

LOSS_COMBOS = [(vl, cv, dc) for vl in ("mse", "huber") for cv in (False, True) for dc in (False, True)]


def gen_loss():
    T, B = 24, 6
    rng = np.random.default_rng(11)
    shape = (T, B, 1)
    new_lp = (np.log(0.5) + 0.3 * rng.standard_normal(shape)).astype(np.float32)
    old_lp = (np.log(0.5) + 0.3 * rng.standard_normal(shape)).astype(np.float32)
    value = rng.standard_normal(shape).astype(np.float32) * 2
    old_value = (value + 0.5 * rng.standard_normal(shape)).astype(np.float32)
    adv = rng.standard_normal(shape).astype(np.float32) * 4
    ret = (old_value + adv + 15 * (rng.random(shape) < 0.1)).astype(np.float32)  # a few beyond huber delta
    entropy = (0.6 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
    mask = (rng.random(shape) < 0.85).astype(np.float32)
    out = dict(new_lp=new_lp, old_lp=old_lp, value=value, old_value=old_value, adv=adv, ret=ret, entropy=entropy,
               mask=mask)
    for vl, cv, dc in LOSS_COMBOS:
        targs = dict(value_loss=vl, clip_value=cv, dual_clip=dc, popart=False,
                     value_loss_config=dict(delta=10.0) if vl == "huber" else {})
        trainer = make_ref_trainer(C1_POLICY, targs)
        tl = lambda a, g=False: torch.from_numpy(a).clone().requires_grad_(g)
        nlp, v, ent = tl(new_lp, True), tl(value, True), tl(entropy, True)
        sample = NamedArray(on_reset=torch.zeros(shape), done=torch.zeros(shape), truncated=torch.zeros(shape),
                            analyzed_result=NamedArray(adv=tl(adv), ret=tl(ret), value=tl(old_value)))
        ar = SampleAnalyzedResult(old_action_log_probs=tl(old_lp), new_action_log_probs=nlp, state_values=v,
                                  entropy=ent)
        loss, res = trainer._compute_loss(sample, ar, tl(mask))
        loss.backward()
        tag = f"{vl}_{int(cv)}_{int(dc)}"
        stats = dict(loss=loss.item(), policy_loss=res.policy_loss.item(), value_loss=res.value_loss.item(),
                     entropy=res.entropy.item(), clip_ratio=res.clip_ratio.mean().item(),
                     importance_weight=res.importance_weight.mean().item(), advantage=res.advantage.mean().item(),
                     value_targets=res.value_targets.mean().item())
        # oracle cross-check
        o_nlp, o_v, o_ent = tl(new_lp, True), tl(value, True), tl(entropy, True)
        o_loss, o_stats = oppo.ppo_loss(o_nlp, tl(old_lp), o_v, tl(old_value), tl(adv), tl(ret), o_ent, tl(mask),
                                        dual_clip=dc, value_loss=vl, clip_value=cv,
                                        value_loss_config=targs["value_loss_config"])
        o_loss.backward()
        assert abs(o_loss.item() - loss.item()) <= 1e-6 * max(1, abs(loss.item())), tag
        for a, b_ in [(o_nlp.grad, nlp.grad), (o_v.grad, v.grad), (o_ent.grad, ent.grad)]:
            assert torch.allclose(a, b_, rtol=1e-6, atol=1e-9), tag
        for k, val in stats.items():
            if k != "loss":
                assert abs(o_stats[k] - val) <= 1e-6 * max(1, abs(val)), (tag, k)
        out[f"{tag}_stats"] = np.array([stats[k] for k in sorted(stats)], dtype=np.float64)
        out[f"{tag}_d_new_lp"], out[f"{tag}_d_value"], out[f"{tag}_d_entropy"] = (nlp.grad.numpy(), v.grad.numpy(),
                                                                                   ent.grad.numpy())
    out["stat_names"] = np.array(sorted(stats))
    out["combos"] = np.array([f"{vl}_{int(cv)}_{int(dc)}" for vl, cv, dc in LOSS_COMBOS])
    save("loss.npz", **out)


# ------------------------------------------------------------------------------------------------ G5/G6/G7
def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def check_init(policy_args, ref_sd):
    """This repo's initialiser must reproduce the reference's initial weights bit for bit."""
    args = {k: v for k, v in policy_args.items() if k != "chunk_len"}
    spec, vals = build_netspec(**args)
    assert list(vals.keys()) == list(ref_sd.keys()), "state_dict key order differs"
    for k in vals:
        assert torch.equal(vals[k], ref_sd[k]), f"init mismatch at {k}"
    return spec


def run_steps(tag, policy_args, trainer_args, sample_kw, n_steps, store_state="full", out=None, analyze_check=True):
    out = {} if out is None else out
    trainer = make_ref_trainer(policy_args, trainer_args)
    net = trainer.policy.net
    sd0 = sd_to_np(net.state_dict())
    check_init(policy_args, net.state_dict())
    out[f"{tag}_init_sha"] = np.array(state_sha(net.state_dict()))
    oracle_net = OracleActorCritic(**policy_args)
    oracle_net.load_state_dict(sd0)
    oracle = OracleMappo(oracle_net, **trainer_args)
    stat_keys = None
    for step in range(n_steps):
        arrays = synthetic.make_sample_arrays(seed=100 + step, **sample_kw)
        sample = ref_sample({k: v.copy() for k, v in arrays.items()})
        if step == 0 and analyze_check:
            # G6: analysis outputs on the initial weights
            ts = recursive_apply(sample, lambda x: torch.from_numpy(x).float())
            Tb = arrays["on_reset"].shape[0]
            with torch.no_grad():
                ar = trainer.policy.analyze(ts[:Tb - 1], target="ppo", burn_in_steps=0)
                pnames = ["policy_state.hx"] if "policy_state.hx" in arrays else \
                    [n for n in ("policy_state.actor_hx", "policy_state.critic_hx") if n in arrays]
                lp, v, ent, _ = oracle_net.analyze({k[4:]: torch.from_numpy(a[:Tb - 1]).float()
                                                    for k, a in arrays.items() if k.startswith("obs.")},
                                                   torch.from_numpy(arrays["action.x"][:Tb - 1]).float(),
                                                   torch.from_numpy(arrays["on_reset"][:Tb - 1]).float(),
                                                   [torch.from_numpy(arrays[n][:Tb - 1]) for n in pnames] or None)
            assert torch.allclose(lp, ar.new_action_log_probs, rtol=1e-5, atol=1e-6)
            assert torch.allclose(v, ar.state_values, rtol=1e-5, atol=1e-6)
            assert torch.allclose(ent, ar.entropy, rtol=1e-5, atol=1e-6)
            out[f"{tag}_analyze_new_lp"] = ar.new_action_log_probs.numpy()
            out[f"{tag}_analyze_value"] = ar.state_values.numpy()
            out[f"{tag}_analyze_entropy"] = ar.entropy.numpy()
        res = trainer.step(sample)
        o_stats, o_out = oracle.step(arrays)
        stats = {k: float(v) for k, v in res.stats.items()}
        for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "clip_ratio", "importance_weight", "advantage",
                  "value_targets", "done", "truncated"):
            assert abs(o_stats[k] - stats[k]) <= 2e-5 * max(1.0, abs(stats[k])), (tag, step, k, o_stats[k], stats[k])
        if trainer_args.get("vtrace"):  # the ratio comes out of the network: equal to float32 rounding, not bitwise
            assert np.allclose(o_out["adv"], sample.analyzed_result.adv, rtol=1e-5, atol=1e-6)
            assert np.allclose(o_out["ret"], sample.analyzed_result.ret, rtol=1e-5, atol=1e-6)
        else:
            assert np.array_equal(o_out["adv"], sample.analyzed_result.adv)
            assert np.array_equal(o_out["ret"], sample.analyzed_result.ret)
        stat_keys = sorted(stats)
        out[f"{tag}_step{step}_stats"] = np.array([stats[k] for k in stat_keys], dtype=np.float64)
        if step == 0:
            out[f"{tag}_step0_adv"], out[f"{tag}_step0_ret"] = sample.analyzed_result.adv, sample.analyzed_result.ret
        sd = sd_to_np(net.state_dict())
        osd = oracle_net.state_dict()
        for k in sd:
            assert np.allclose(osd[k].numpy(), sd[k], rtol=1e-4, atol=1e-6), (tag, step, k)
        if step in (0, n_steps - 1):
            for k, v in sd.items():
                if store_state == "full" or v.size <= 4096:
                    out[f"{tag}_step{step}_param:{k}"] = v
                else:  # large tensors: a strided subsample (stride recorded in the key)
                    out[f"{tag}_step{step}_param_s97:{k}"] = v.reshape(-1)[::97].copy()
    out[f"{tag}_stat_names"] = np.array(stat_keys)
    out[f"{tag}_version"] = np.array(trainer.policy.version)
    if store_state == "full":
        for k, v in sd0.items():
            out[f"{tag}_init_param:{k}"] = v
    return out


def gen_steps():
    c1_sample = dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05)
    out = {}
    # config #1 of BASELINE.json: CartPole shapes, separate 2x64 MLP, reference-default hyper-parameters, lr 3e-4
    run_steps("c1", C1_POLICY, dict(popart=False, optimizer_config=dict(lr=3e-4)), c1_sample, 3, out=out)
    # same shapes with the Atari preset of legacy/experiments/atari.py:952-973
    atari_trainer = dict(discount_rate=0.99, gae_lambda=0.97, eps_clip=0.2, clip_value=True, dual_clip=False,
                         value_loss='huber', value_loss_weight=1.0, value_loss_config=dict(delta=10.0),
                         entropy_bonus_weight=0.01, optimizer='adam', optimizer_config=dict(lr=5e-4), popart=False,
                         max_grad_norm=40.0, bootstrap_steps=1)
    run_steps("c1atari", C1_POLICY, atari_trainer, c1_sample, 2, out=out)
    # MLP variants: layernorm on, shared backbone, multi-discrete heads, two observation keys, tanh
    ln_policy = dict(C1_POLICY, layernorm=True, seed=2)
    run_steps("c1ln", ln_policy, dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=0.5), c1_sample, 2,
              out=out)
    multi_policy = dict(obs_dim={"a": 5, "b": 3}, action_dim=[3, 4], hidden_dim=32, num_dense_layers=1,
                        num_rnn_layers=0, popart=False, layernorm=True, shared_backbone=True, chunk_len=8, seed=3,
                        activation="tanh")
    multi_sample = dict(T=16, B=4, obs_spec={"a": ((5,), "f32"), "b": ((3,), "f32")}, action_dims=[3, 4], p_done=0.1)
    run_steps("multi", multi_policy, dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=1e-3)), multi_sample, 2,
              out=out)
    save("steps_mlp.npz", **out)

    # down-sized Atari: NatureCNN on (4,84,84) uint8 frames, shared backbone, Atari preset
    cnn_policy = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
                      popart=False, layernorm=False, shared_backbone=True, chunk_len=4, seed=5,
                      cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))
    cnn_sample = dict(T=4, B=3, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.1)
    out = run_steps("cnn", cnn_policy, atari_trainer, cnn_sample, 2, store_state="sampled")
    save("steps_cnn.npz", **out)


PAD_POLICY = dict(obs_dim={"obs": (4, 20, 20)}, action_dim=5, hidden_dim=32, num_dense_layers=1, num_rnn_layers=0,
                  popart=False, layernorm=False, shared_backbone=True, chunk_len=4, seed=71,
                  cnn_layers=dict(obs=[(8, 3, 1, 1, 'zeros'), (16, 3, 2, 1, 'zeros'), (8, 3, 1, 0, 'zeros')]))
POOL_POLICY = dict(obs_dim={"img": (3, 23, 19), "vec": 5}, action_dim=[3, 2], hidden_dim=16, num_dense_layers=1,
                   num_rnn_layers=0, popart=True, layernorm=True, shared_backbone=False, chunk_len=4, seed=72,
                   activation="tanh", use_maxpool=dict(img=True),
                   cnn_layers=dict(img=[(4, 3, 1, 0, 'zeros'), (8, 3, 1, 1, 'zeros'), (4, 3, 1, 0, 'zeros')]))


def gen_cnn_padpool():
    """Convolution encoders with zero padding and with nn.MaxPool2d(2) between the layers (modules/cnn.py:99-126: the pooling
    layer sits BEFORE every convolution but the last, so the first one pools the layer-normed image itself).  Odd image
    sizes (floor windows), three input channels, tanh, float32 frames and a vector key beside the image."""
    out = {}
    run_steps("cnnpad", PAD_POLICY, dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=5e-4), max_grad_norm=10.0),
              dict(T=6, B=4, obs_spec={"obs": ((4, 20, 20), "u8")}, action_dims=5, p_done=0.1), 2, out=out, store_state="sampled")
    run_steps("cnnpool", POOL_POLICY, dict(popart=True, optimizer_config=dict(lr=1e-3)),
              dict(T=5, B=3, obs_spec={"img": ((3, 23, 19), "f32"), "vec": ((5,), "f32")}, action_dims=[3, 2], p_done=0.1), 2,
              out=out)
    save("steps_cnn_padpool.npz", **out)


ND1_POLICY = dict(obs_dim={"seq": (3, 59)}, action_dim=4, hidden_dim=16, num_dense_layers=1, num_rnn_layers=0, popart=False,
                  layernorm=True, shared_backbone=True, chunk_len=4, seed=73, use_maxpool=dict(seq=True),
                  cnn_layers=dict(seq=[(4, 3, 1, 0, 'zeros'), (8, 3, 2, 1, 'zeros'), (4, 3, 1, 0, 'zeros')]))
ND3_POLICY = dict(obs_dim={"vol": (2, 9, 8, 7), "vec": 3}, action_dim=[2, 3], hidden_dim=16, num_dense_layers=1,
                  num_rnn_layers=0, popart=False, layernorm=False, shared_backbone=False, chunk_len=4, seed=74,
                  activation="tanh", use_maxpool=dict(vol=True),
                  cnn_layers=dict(vol=[(4, 3, 1, 1, 'zeros'), (4, 2, 1, 0, 'zeros')]))


def gen_cnn_nd():
    """Convolution encoders of observations with one and three spatial dimensions (modules/cnn.py:60-71: nn.Conv1d / nn.Conv3d
    with MaxPool1d / MaxPool3d; the shapes of modules_test.py:385-401 scaled down): pooling, stride, padding, uint8 and
    float32 observations, a vector key beside the volume."""
    out = {}
    run_steps("cnn1d", ND1_POLICY, dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0),
              dict(T=6, B=4, obs_spec={"seq": ((3, 59), "f32")}, action_dims=4, p_done=0.1), 2, out=out)
    run_steps("cnn3d", ND3_POLICY, dict(popart=False, optimizer_config=dict(lr=1e-3)),
              dict(T=5, B=3, obs_spec={"vol": ((2, 9, 8, 7), "u8"), "vec": ((3,), "f32")}, action_dims=[2, 3], p_done=0.1), 2,
              out=out)
    save("steps_cnn_nd.npz", **out)


PADM_POLICY = dict(obs_dim={"img": (4, 12, 10), "seq": (2, 21)}, action_dim=3, hidden_dim=16, num_dense_layers=1,
                   num_rnn_layers=0, popart=False, layernorm=False, shared_backbone=True, chunk_len=4, seed=75,
                   cnn_layers=dict(img=[(8, 3, 1, 2, 'reflect'), (8, 3, 2, 1, 'circular'), (4, 3, 1, 1, 'zeros')],
                                   seq=[(4, 5, 2, 3, 'replicate'), (4, 3, 1, 1, 'circular')]))


def gen_cnn_padmode():
    """nn.ConvNd's non-zero padding modes (modules/cnn.py:107-113 passes `padding_mode` through): reflect, circular and
    replicate borders, with strides, on a uint8 image and a float32 sequence in one policy."""
    out = {}
    run_steps("padm", PADM_POLICY, dict(popart=False, ppo_epochs=2, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0),
              dict(T=5, B=4, obs_spec={"img": ((4, 12, 10), "u8"), "seq": ((2, 21), "f32")}, action_dims=3, p_done=0.1), 2,
              out=out)
    save("steps_cnn_padmode.npz", **out)


def gen_vtrace_rnn():
    """V-trace with recurrent policies (mappo.py:243-246: the analysed rows give both the importance ratio of the trace
    and the loss): GRU shared backbone, and LSTM separate backbones with PopArt and two epochs."""
    out = {}
    vt_gru = dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=1, popart=False, layernorm=True,
                  shared_backbone=True, chunk_len=8, seed=51)
    smp = dict(T=32, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.08, p_trunc=0.0, policy_state={"hx": (1, 32)})
    run_steps("vtgru", vt_gru, dict(popart=False, vtrace=True, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0), smp, 2,
              out=out)
    vt_lstm = dict(obs_dim=4, action_dim=[3, 2], hidden_dim=16, num_dense_layers=1, num_rnn_layers=1, rnn_type="lstm",
                   popart=True, layernorm=False, shared_backbone=False, chunk_len=4, seed=52)
    smp2 = dict(T=16, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1, p_trunc=0.0,
                policy_state={"actor_hx": (1, 32), "critic_hx": (1, 32)})
    run_steps("vtlstm", vt_lstm, dict(popart=True, vtrace=True, ppo_epochs=2, optimizer_config=dict(lr=5e-4)), smp2, 2, out=out)
    # bootstrap_steps = 2 under V-trace (feed-forward): the first epoch analyses Tb - 1 rows for the ratio, the loss takes Tb - 2
    run_steps("vtb2", dict(C1_POLICY, chunk_len=1, seed=53), dict(popart=False, vtrace=True, bootstrap_steps=2, ppo_epochs=2,
                                                     optimizer_config=dict(lr=1e-3)),
              dict(T=31, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05, p_trunc=0.0, bootstrap_steps=2), 2,
              out=out, analyze_check=False)
    save("steps_vtrace_rnn.npz", **out)


def _strip_asserts(fn):
    """The function recompiled without its `assert` statements: what `python -O` -- which the reference's launcher uses
    (apps/main.py:45) -- executes.  Harness-side only; the reference's source is read, not modified."""
    import ast
    import inspect
    import textwrap
    fn = inspect.unwrap(fn)  # e.g. the function under @torch.no_grad(): the decorator line is in the source and re-applies
    tree = ast.parse(textwrap.dedent(inspect.getsource(fn)))

    class Drop(ast.NodeTransformer):

        def visit_Assert(self, node):
            return ast.Pass()

    tree = ast.fix_missing_locations(Drop().visit(tree))
    ns = dict(fn.__globals__)
    exec(compile(tree, inspect.getsourcefile(fn), "exec"), ns)
    return ns[fn.__name__]


def gen_value_dim():
    """value_dim > 1 through the whole step (mappo.py:146-217 is shape-agnostic: ratio, mask and entropy broadcast over the
    value channels).  The reference's masked_normalization ASSERTS mask.shape == x.shape (utils.py:48-53), which a [T, B, 1]
    loss mask fails for [T, B, value_dim] advantages -- so this configuration only runs the way the launcher runs it, under
    `python -O`: the generator swaps in that function with its asserts stripped."""
    import legacy.algorithm.modules.utils as ref_utils
    saved = modules.masked_normalization
    modules.masked_normalization = ref_utils.masked_normalization = _strip_asserts(saved)
    try:
        out = {}
        smp = dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05, value_dim=3)
        run_steps("vd3", dict(C1_POLICY, value_dim=3, seed=61), dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=5.0),
                  smp, 2, out=out)
        smp2 = dict(T=16, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1, value_dim=2)
        run_steps("vd2pa", dict(C1_POLICY, action_dim=[3, 2], value_dim=2, popart=True, layernorm=True, shared_backbone=True,
                                seed=62),
                  dict(popart=True, ppo_epochs=2, clip_value=True, dual_clip=False, value_loss='huber',
                       value_loss_config=dict(delta=10.0), optimizer_config=dict(lr=5e-4)), smp2, 2, out=out)
        save("steps_value_dim.npz", **out)
    finally:
        modules.masked_normalization = ref_utils.masked_normalization = saved


def gen_optim():
    """The other optimisers modules/utils.py:268-286 accepts: RMSprop (plain; centred with momentum and weight decay) and
    SGD (plain; Nesterov momentum with weight decay; momentum with dampening), full steps on the CartPole shapes."""
    c1_sample = dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05)
    out = {}
    run_steps("rms", dict(C1_POLICY, seed=41), dict(popart=False, optimizer='rmsprop', optimizer_config=dict(lr=1e-3)),
              c1_sample, 3, out=out, analyze_check=False)
    run_steps("rmsc", dict(C1_POLICY, layernorm=True, seed=42),
              dict(popart=False, optimizer='rmsprop', max_grad_norm=5.0,
                   optimizer_config=dict(lr=5e-4, alpha=0.95, eps=1e-6, momentum=0.9, centered=True, weight_decay=1e-3)),
              c1_sample, 3, out=out, analyze_check=False)
    run_steps("sgd", dict(C1_POLICY, seed=43), dict(popart=False, optimizer='sgd', optimizer_config=dict(lr=1e-2)), c1_sample,
              2, out=out, analyze_check=False)
    run_steps("sgdn", dict(C1_POLICY, shared_backbone=True, seed=44),
              dict(popart=False, optimizer='sgd', max_grad_norm=1.0,
                   optimizer_config=dict(lr=1e-2, momentum=0.9, nesterov=True, weight_decay=1e-3)), c1_sample, 3, out=out,
              analyze_check=False)
    run_steps("sgdd", dict(C1_POLICY, seed=45),
              dict(popart=False, optimizer='sgd', ppo_epochs=2, optimizer_config=dict(lr=1e-2, momentum=0.8, dampening=0.3)),
              c1_sample, 2, out=out, analyze_check=False)
    save("steps_optim.npz", **out)


def gen_popart():
    """PopArt value head (the reference policy's default) and V-trace through the trainer: full steps."""
    c1_sample = dict(T=32, B=8, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05)
    out = {}
    pa_policy = dict(C1_POLICY, popart=True, seed=7)
    run_steps("pa", pa_policy, dict(popart=True, optimizer_config=dict(lr=3e-4)), c1_sample, 3, out=out)
    # shared backbone + layernorm, Atari-style loss (huber, clipped value), two epochs
    pa2_policy = dict(C1_POLICY, popart=True, layernorm=True, shared_backbone=True, seed=8)
    run_steps("pa2", pa2_policy, dict(popart=True, clip_value=True, dual_clip=False, value_loss='huber',
                                     value_loss_config=dict(delta=10.0), value_loss_weight=1.0, ppo_epochs=2,
                                     optimizer_config=dict(lr=5e-4), max_grad_norm=40.0), c1_sample, 2, out=out)
    # V-trace (no PopArt), and both together.  No truncated steps in these samples: one of the reference's debug
    # assertions (gae.py:72, stripped by the launcher's `python -O`) does not hold for truncations under V-trace.
    vt_sample = dict(c1_sample, p_trunc=0.0)
    run_steps("vt", dict(C1_POLICY, seed=9), dict(popart=False, vtrace=True, optimizer_config=dict(lr=3e-4)), vt_sample, 2,
              out=out)
    run_steps("vtpa", dict(C1_POLICY, popart=True, seed=10), dict(popart=True, vtrace=True, max_grad_norm=10.0,
                                                                 optimizer_config=dict(lr=1e-3)), vt_sample, 2, out=out)
    save("steps_popart.npz", **out)


def gen_rnn():
    """Recurrent backbones (GRU + auto reset, chunked analysis from stored states): full steps and a stateful rollout."""
    out = {}
    sh_policy = dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=1, popart=False,
                     layernorm=True, shared_backbone=True, chunk_len=8, seed=21)
    sh_sample = dict(T=32, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.08, policy_state={"hx": (1, 32)})
    run_steps("gru", sh_policy, dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=10.0), sh_sample, 3,
              out=out)
    # separate backbones, two GRU layers, no LayerNorm in the dense stack (ReLU output feeds the GRU), PopArt head,
    # the chunk is the whole trajectory
    sep_policy = dict(obs_dim=4, action_dim=[3, 2], hidden_dim=16, num_dense_layers=2, num_rnn_layers=2, popart=True,
                      layernorm=False, shared_backbone=False, chunk_len=16, seed=22)
    sep_sample = dict(T=16, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1,
                      policy_state={"actor_hx": (2, 16), "critic_hx": (2, 16)})
    run_steps("gru2", sep_policy, dict(popart=True, ppo_epochs=2, optimizer_config=dict(lr=5e-4)), sep_sample, 2, out=out)
    # LSTM cell: the stored state is cat(h, c); shared backbone, two layers, chunks of 4
    lstm_policy = dict(obs_dim=4, action_dim=2, hidden_dim=16, num_dense_layers=1, num_rnn_layers=2, rnn_type="lstm",
                       popart=False, layernorm=True, shared_backbone=True, chunk_len=4, seed=23)
    lstm_sample = dict(T=16, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.1, policy_state={"hx": (2, 32)})
    run_steps("lstm", lstm_policy, dict(popart=False, optimizer_config=dict(lr=1e-3)), lstm_sample, 2, out=out)
    # burn-in: the 2 rows before every chunk are replayed without gradient to produce the chunk's initial state
    # (18 stored rows + bootstrap: 2 burn-in + 16 = 4 chunks of 4)
    bi_policy = dict(obs_dim=4, action_dim=2, hidden_dim=16, num_dense_layers=1, num_rnn_layers=1, popart=False,
                     layernorm=True, shared_backbone=False, chunk_len=4, seed=24)
    bi_sample = dict(T=18, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.1,
                     policy_state={"actor_hx": (1, 16), "critic_hx": (1, 16)})
    run_steps("burn", bi_policy, dict(popart=False, burn_in_steps=2, optimizer_config=dict(lr=1e-3)), bi_sample, 2, out=out,
              analyze_check=False)
    # stateful deterministic rollout
    policy = api.policy.make(api.config.Policy("actor-critic", args=sh_policy))
    policy.eval_mode()
    rng = np.random.default_rng(6)
    N = 9
    obs = rng.standard_normal((N, 4)).astype(np.float32)
    hx = (0.5 * rng.standard_normal((N, 1, 32))).astype(np.float32)
    req = api.policy.RolloutRequest(obs=NamedArray(obs=obs), policy_state=NamedArray(hx=hx),
                                    is_evaluation=np.ones((N, 1), dtype=np.uint8), on_reset=np.zeros((N, 1), dtype=np.uint8))
    res = policy.rollout(req)
    out["roll_obs"], out["roll_hx"] = obs, hx
    out["roll_action"], out["roll_log_probs"], out["roll_value"] = (res.action.x, res.analyzed_result.log_probs,
                                                                    res.analyzed_result.value)
    out["roll_new_hx"] = res.policy_state.hx
    for k, v in sd_to_np(policy.net.state_dict()).items():
        out[f"roll_param:{k}"] = v
    save("steps_rnn.npz", **out)


SMAC_3M = ((30,), (48,), 9, 3)  # (obs, state, #actions, #agents): the standard 3m feature sizes


def gen_smac():
    """``smac_rnn`` (game_policies/smac_rnn.py), shared agents: full trainer steps on ``[Tb, B, agents, ...]`` samples,
    and a rollout.

    Harness-side shims (the reference file itself is untouched): (1) ``legacy.environment.smac.smac_env`` needs a
    StarCraft installation; a stand-in module supplies ``SMACAction`` and ``get_smac_shapes`` returning the 3m sizes.
    (2) The module is imported explicitly (``legacy/algorithm/ppo/game_policies/__init__.py`` has the import commented
    out).  (3) For the rollout only: ``SMACNet`` gets ``AutoResetRNN``'s default LSTM, whose state is 2H wide, while
    ``SMACPolicy``'s default state is H wide, so the stock rollout raises inside ``nn.LSTM``; the private default is
    replaced by a 2H-wide zero array before calling it.  The training side needs no such patch."""
    import types
    from oracle.net import OracleSMACNet
    from srl_amd.algorithm.netspec import build_smac_netspec
    fake = types.ModuleType("legacy.environment.smac.smac_env")

    class SMACAction(DiscreteAction):
        pass

    fake.SMACAction = SMACAction
    fake.get_smac_shapes = lambda map_name, **kw: SMAC_3M
    sys.modules["legacy.environment.smac.smac_env"] = fake
    import legacy.algorithm.ppo.game_policies.smac_rnn  # noqa: F401  (registers "smac_rnn")

    out = {}
    H, A, CL = 32, 3, 5
    policy_args = dict(map_name="3m", hidden_dim=H, chunk_len=CL, seed=31, shared=True)
    trainer_args = dict(popart=True, ppo_epochs=2, optimizer_config=dict(lr=5e-4, eps=1e-5), max_grad_norm=10.0,
                        value_loss="huber", value_loss_config=dict(delta=10.0), clip_value=True, dual_clip=False)
    trainer = make_ref_trainer(policy_args, trainer_args, "smac_rnn")
    net = trainer.policy.net
    sd0 = sd_to_np(net.state_dict())
    spec, vals = build_smac_netspec(30, 48, 9, H, seed=31)
    assert list(vals.keys()) == list(sd0.keys()), "state_dict key order differs"
    for k in vals:
        assert torch.equal(vals[k], net.state_dict()[k]), f"init mismatch at {k}"
    oracle_net = OracleSMACNet(30, 48, 9, H, CL)
    oracle_net.load_state_dict(sd0)
    oracle = OracleMappo(oracle_net, **trainer_args)
    sample_kw = dict(T=20, B=4, agents=A, obs_spec={"local_obs": ((30,), "f32"), "state": ((48,), "f32")}, action_dim=9,
                     p_done=0.08, policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
    n_steps = 2
    for step in range(n_steps):
        arrays = synthetic.make_multiagent_arrays(seed=300 + step, **sample_kw)
        sample = ref_sample({k: v.copy() for k, v in arrays.items()})
        if step == 0:
            ts = recursive_apply(sample, lambda x: torch.from_numpy(x).float())
            Tb = arrays["on_reset"].shape[0]
            with torch.no_grad():
                ar = trainer.policy.analyze(ts[:Tb - 1], target="ppo", burn_in_steps=0)
                lp, v, ent, _ = oracle_net.analyze({k[4:]: torch.from_numpy(a[:Tb - 1]).float()
                                                    for k, a in arrays.items() if k.startswith("obs.")},
                                                   torch.from_numpy(arrays["action.x"][:Tb - 1]).float(),
                                                   torch.from_numpy(arrays["on_reset"][:Tb - 1]).float(),
                                                   [torch.from_numpy(arrays[n][:Tb - 1]) for n in
                                                    ("policy_state.actor_hx", "policy_state.critic_hx")])
            assert torch.equal(torch.isinf(lp), torch.isinf(ar.new_action_log_probs))
            fin = torch.isfinite(lp)
            assert torch.allclose(lp[fin], ar.new_action_log_probs[fin], rtol=1e-5, atol=1e-6)
            assert torch.allclose(v, ar.state_values, rtol=1e-5, atol=1e-6)
            assert torch.allclose(ent, ar.entropy, rtol=1e-5, atol=1e-6)
            out["smac_analyze_new_lp"] = ar.new_action_log_probs.numpy()
            out["smac_analyze_value"] = ar.state_values.numpy()
            out["smac_analyze_entropy"] = ar.entropy.numpy()
        res = trainer.step(sample)
        o_stats, o_out = oracle.step(arrays)
        stats = {k: float(v) for k, v in res.stats.items()}
        for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "clip_ratio", "importance_weight", "advantage",
                  "value_targets", "done", "truncated", "denorm_value"):
            assert abs(o_stats[k] - stats[k]) <= 2e-5 * max(1.0, abs(stats[k])), ("smac", step, k, o_stats[k], stats[k])
        assert np.array_equal(o_out["adv"], sample.analyzed_result.adv)
        assert np.array_equal(o_out["ret"], sample.analyzed_result.ret)
        stat_keys = sorted(stats)
        out[f"smac_step{step}_stats"] = np.array([stats[k] for k in stat_keys], dtype=np.float64)
        if step == 0:
            out["smac_step0_adv"], out["smac_step0_ret"] = sample.analyzed_result.adv, sample.analyzed_result.ret
        sd = sd_to_np(net.state_dict())
        osd = oracle_net.state_dict()
        for k in sd:
            assert np.allclose(osd[k].numpy(), sd[k], rtol=1e-4, atol=1e-6), ("smac", step, k)
        for k, v in sd.items():
            out[f"smac_step{step}_param:{k}"] = v
    out["smac_stat_names"] = np.array(stat_keys)
    out["smac_version"] = np.array(trainer.policy.version)
    out["smac_init_sha"] = np.array(state_sha(OrderedDictNP(sd0)))
    for k, v in sd0.items():
        out[f"smac_init_param:{k}"] = v

    # deterministic rollout of [N, agents, ...] requests on the trained weights, carried states on some rows reset
    policy = trainer.policy
    setattr(policy, "_SMACPolicy__rnn_default_hidden", np.zeros((A, 1, 2 * H), dtype=np.float32))  # shim (3)
    rng = np.random.default_rng(12)
    N = 5
    avail = (rng.random((N, A, 9)) < 0.6).astype(np.uint8)
    avail[..., 0] = 1
    req = dict(local_obs=rng.standard_normal((N, A, 30)).astype(np.float32),
               state=rng.standard_normal((N, A, 48)).astype(np.float32), available_action=avail,
               is_alive=np.ones((N, A, 1), dtype=np.uint8))
    hx = (0.5 * rng.standard_normal((2, N, A, 1, 2 * H))).astype(np.float32)
    on_reset = (rng.random((N, 1, 1)) < 0.4).astype(np.uint8).repeat(A, axis=1)
    aux = {k: np.zeros((N, A), dtype=np.int32) for k in ("client_id", "request_id", "received_time", "buffer_index",
                                                          "step_count", "ready")}
    r = api.policy.RolloutRequest(obs=NamedArray(**req), policy_state=NamedArray(actor_hx=hx[0], critic_hx=hx[1]),
                                  is_evaluation=np.ones((N, A, 1), dtype=np.uint8), on_reset=on_reset, **aux)
    res = policy.rollout(r)
    for k, v in req.items():
        out[f"smac_roll_obs.{k}"] = v
    out["smac_roll_actor_hx"], out["smac_roll_critic_hx"], out["smac_roll_on_reset"] = hx[0], hx[1], on_reset
    out["smac_roll_action"], out["smac_roll_log_probs"] = res.action.x, res.analyzed_result.log_probs
    out["smac_roll_value"] = res.analyzed_result.value
    out["smac_roll_new_actor_hx"], out["smac_roll_new_critic_hx"] = res.policy_state.actor_hx, res.policy_state.critic_hx
    save("steps_smac.npz", **out)


def gen_presets():
    """Per-game presets of ActorCriticPolicy (football, atari-vision, overcooked): parameter tables (state_dict keys and
    shapes) and a strided subsample of the initial weights for a fixed seed.  (football-smm's default convolution stack
    ends in a 22528 -> 11264 Linear, 254 M weights: its table is recorded from a meta-device construction of the same
    module tree, without values.)"""
    from srl_amd.algorithm import game_policies as gp
    out = {}
    for name in ("football-simple115-separate", "overcooked-separate", "atari-vision"):
        policy = api.policy.make(api.config.Policy(name, args=dict(seed=41)))
        sd = policy.net.state_dict()
        out[f"{name}:keys"] = np.array(list(sd.keys()))
        out[f"{name}:shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
        for k, v in sd.items():
            stride = 97 if v.numel() < 100000 else 4999
            out[f"{name}:init_s{stride}:{k}"] = v.detach().numpy().reshape(-1)[::stride].copy()
        cls = api.policy.ALL_POLICY_CLASSES[name] if hasattr(api.policy, "ALL_POLICY_CLASSES") else None
        mine = {"football-simple115-separate": gp.FootballSeparatePolicy, "overcooked-separate": gp.OvercookedSeparatePolicy,
                "atari-vision": gp.AtariVisionPolicy}[name]
        args = {k: v for k, v in dict(mine.defaults, seed=41).items() if k != "chunk_len"}
        spec, vals = build_netspec(**args)
        assert list(vals.keys()) == list(sd.keys()), name
        for k in vals:
            assert torch.equal(vals[k], sd[k]), (name, k)
    # football-smm: names and shapes only
    from legacy.algorithm.ppo.actor_critic_policies.actor_critic_policy import ActorCriticSeparate
    with torch.device("meta"):
        net = ActorCriticSeparate(obs_dim={"obs": (4, 96, 72)}, action_dim=19, hidden_dim=128, value_dim=1, state_dim=None,
                                  cnn_layers={}, use_maxpool={}, dense_layers=2, rnn_type="gru", num_rnn_layers=1,
                                  popart=True, activation="relu", layernorm=True, shared_backbone=False,
                                  continuous_action=False, auxiliary_head=False)
    sd = net.state_dict()
    out["football-smm-separate:keys"] = np.array(list(sd.keys()))
    out["football-smm-separate:shapes"] = np.array([",".join(map(str, v.shape)) for v in sd.values()])
    save("presets.npz", **out)


class OrderedDictNP(dict):
    """state_sha wants tensors; wrap numpy arrays."""

    def items(self):
        return [(k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in super().items()]


def gen_continuous():
    """Continuous actions (Normal(mean, std), the `gym_mujoco` policy): the three std parametrisations, full steps and a
    deterministic rollout."""
    out = {}
    base = dict(obs_dim=7, action_dim=3, hidden_dim=32, num_dense_layers=2, num_rnn_layers=0, popart=False,
                layernorm=True, shared_backbone=False, chunk_len=8, continuous_action=True)
    smp = dict(T=16, B=6, obs_spec={"obs": ((7,), "f32")}, action_dims=3, p_done=0.1, continuous_action=True)
    tr = dict(popart=False, optimizer_config=dict(lr=1e-3), max_grad_norm=5.0)
    run_steps("cfix", dict(base, std_type="fixed", init_log_std=-0.3, seed=31), tr, smp, 2, out=out)
    run_steps("csep", dict(base, std_type="separate_learnable", seed=32), dict(tr, ppo_epochs=2), smp, 2, out=out)
    run_steps("cshr", dict(base, std_type="shared_learnable", shared_backbone=True, seed=33), tr, smp, 2, out=out)
    policy = api.policy.make(api.config.Policy("actor-critic", args=dict(base, std_type="separate_learnable", seed=32)))
    policy.eval_mode()
    rng = np.random.default_rng(8)
    N = 7
    obs = rng.standard_normal((N, 7)).astype(np.float32)
    req = api.policy.RolloutRequest(obs=NamedArray(obs=obs), is_evaluation=np.ones((N, 1), dtype=np.uint8),
                                    on_reset=np.zeros((N, 1), dtype=np.uint8))
    res = policy.rollout(req)
    out["roll_obs"] = obs
    out["roll_action"], out["roll_log_probs"], out["roll_value"] = (res.action.x, res.analyzed_result.log_probs,
                                                                    res.analyzed_result.value)
    for k, v in sd_to_np(policy.net.state_dict()).items():
        out[f"roll_param:{k}"] = v
    save("steps_continuous.npz", **out)


def gen_paramdb():
    """Cross-check of the filesystem parameter store with the reference's client on the same directory: each reads
    what the other wrote; the resulting listing is the fixture."""
    import tempfile
    sys.modules.setdefault("pymongo", mock.MagicMock())
    sys.modules.setdefault("pymongo.errors", mock.MagicMock())
    import distributed.system.parameter_db as ref_db
    import base.names
    from srl_amd.runtime.parameter_db import FilesystemParameterDB
    with tempfile.TemporaryDirectory() as root:
        ref_db.PytorchFilesystemParameterDB.ROOT = root
        ref = ref_db.PytorchFilesystemParameterDB("exp", "trial", user_namespace="ns")
        ours = FilesystemParameterDB("exp", "trial", root=root, user_namespace="ns")
        ck = lambda steps: {"steps": steps, "state_dict": {"w": torch.full((3,), float(steps))}}
        ours.push("pol", ck(5), version="5")
        ref.push("pol", ck(12), version="12", tags="best")
        ours.push("pol", ck(20), version="20", tags=["eval"])
        ref.tag("pol", "5", "first")
        for v in (30, 40, 50):
            ours.push("pol", ck(v), version=str(v))
        assert ref.get("pol")["steps"] == 50 and ours.get("pol", "best")["steps"] == 12
        assert ref.get("pol", "eval")["state_dict"]["w"][0] == 20 and ours.version_of("pol", "first") == 5
        assert ref.list_versions("pol") == ours.list_versions("pol") and sorted(ref.list_tags("pol")) == sorted(
            ours.list_tags("pol"))
        ours.gc("pol", max_untagged_version_count=1)
        ref2 = sorted(ref.list_versions("pol"))
        assert ref2 == sorted(ours.list_versions("pol"))
        save("paramdb.npz", versions_after_gc=np.array(ref.list_versions("pol")),
             tags=np.array(sorted(f"{t}={v}" for t, v in ref.list_tags("pol"))), names=np.array(ref.list_names()))


def gen_rollout():
    out = {}
    for tag, pargs, obs_spec in [("c1", C1_POLICY, synthetic.CARTPOLE_OBS)]:
        policy = api.policy.make(api.config.Policy("actor-critic", args=pargs))
        policy.eval_mode()
        rng = np.random.default_rng(5)
        N = 17
        obs = {k: rng.standard_normal((N, *shape)).astype(np.float32) for k, (shape, _) in obs_spec.items()}
        req = api.policy.RolloutRequest(obs=NamedArray(**obs), is_evaluation=np.ones((N, 1), dtype=np.uint8),
                                        on_reset=np.zeros((N, 1), dtype=np.uint8))
        res = policy.rollout(req)
        out[f"{tag}_obs"] = obs["obs"]
        out[f"{tag}_action"], out[f"{tag}_log_probs"], out[f"{tag}_value"] = (res.action.x,
                                                                             res.analyzed_result.log_probs,
                                                                             res.analyzed_result.value)
        onet = OracleActorCritic(**pargs)
        onet.load_state_dict(sd_to_np(policy.net.state_dict()))
        a, lp, v, _, _ = onet.rollout_eval({k: torch.from_numpy(x) for k, x in obs.items()})
        assert np.array_equal(a.numpy(), res.action.x) and np.allclose(lp.numpy(), res.analyzed_result.log_probs,
                                                                       atol=1e-6)
    save("rollout.npz", **out)


# ------------------------------------------------------------------------------------------------ G8/G9/G10
def gen_host():
    out = {}
    # G8: TrajGAE on the two hand cases of legacy/tests/modules_test.py:140-178
    for tag, rew, value, done, trunc in [("trunc", [1, 2, 0], [2, 0, 1], [0, 0, 0], [0, 0, 1]),
                                         ("done", [1, 3, 0], [2, 2, 0], [0, 0, 1], [0, 0, 0])]:
        memory = [
            api.trainer.SampleBatch(obs=None, reward=np.array([r], dtype=np.float32),
                                    analyzed_result=PPORolloutAnalyzedResult(value=np.array([v], dtype=np.float32),
                                                                             log_probs=None),
                                    done=np.array([d]), truncated=np.array([t]))
            for r, v, d, t in zip(rew, value, done, trunc)
        ]
        proc = api.trainer.make_traj_postprocessor(api.config.TrajPostprocessor('gae', args=dict(gamma=0.1, lmbda=0.1)))
        memory = proc.process(memory)
        out[f"trajgae_{tag}_in"] = np.array([rew, value, done, trunc], dtype=np.float32)
        out[f"trajgae_{tag}_adv"] = np.array([m.analyzed_result.adv.item() for m in memory[:-1]])
        out[f"trajgae_{tag}_ret"] = np.array([m.analyzed_result.ret.item() for m in memory[:-1]])
    # G10: wire formats
    rng = np.random.default_rng(1)
    na = NamedArray(a=rng.standard_normal((3, 2)).astype(np.float32),
                    b=NamedArray(c=rng.integers(0, 255, (3, 4), dtype=np.uint8), d=None),
                    e=np.arange(3, dtype=np.int64))
    na.register_metadata(tag="golden")
    for method in ("pickle_dict", "raw_bytes"):
        chunks = ref_namedarray.dumps(na, method=method)
        out[f"wire_{method}_n"] = np.array(len(chunks))
        for i, ch in enumerate(chunks):
            out[f"wire_{method}_{i}"] = np.frombuffer(ch, dtype=np.uint8)
    out["wire_a"], out["wire_c"], out["wire_e"] = na.a, na.b.c, na.e
    # G9: PriorityQueueBuffer stacking (base/buffer.py:109-130): batch_size [T,...] samples -> [T,B,...]
    from base.buffer import PriorityQueueBuffer
    buf = PriorityQueueBuffer(max_size=4, reuses=1, batch_size=3)
    singles = []
    for i in range(3):
        arr = synthetic.make_sample_arrays(seed=40 + i, T=4, B=1, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2)
        arr = {k: v[:, 0] for k, v in arr.items()}  # one agent's [Tb, ...] sample
        singles.append(arr)
        buf.put(ref_sample(arr))
    batch = buf.get().sample
    out["buffer_obs"] = batch.obs.obs
    out["buffer_reward"] = batch.reward
    out["buffer_action"] = batch.action.x
    for i, arr in enumerate(singles):
        out[f"buffer_in{i}_obs"], out[f"buffer_in{i}_reward"], out[f"buffer_in{i}_action"] = (arr["obs.obs"],
                                                                                            arr["reward"],
                                                                                            arr["action.x"])
    save("host.npz", **out)


# ------------------------------------------------------------------------------------------------ PPG (second half)
PPG_CASES = {
    # multi-discrete MLP, LayerNorm; one case with an availability mask in the observation (-1e10 logits, :135-136)
    "aux": (dict(obs_dim=4, action_dim=[3, 2], hidden_dim=32, num_dense_layers=1, num_rnn_layers=0, popart=False, layernorm=True,
                 chunk_len=8, seed=81),
            dict(popart=False, ppg_epochs=3, max_grad_norm=5.0, beta_clone=1.0, aux_value_head_weight=0.5,
                 ppg_optimizer_config=dict(lr=1e-3)),
            dict(T=16, B=6, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2], p_done=0.1)),
    "auxmask": (dict(obs_dim=5, action_dim=4, hidden_dim=16, num_dense_layers=2, num_rnn_layers=0,
                     popart=False, layernorm=False, chunk_len=4, seed=82, activation="tanh"),
                dict(popart=False, ppg_epochs=2, beta_clone=2.0, aux_value_head_weight=1.0, ppg_optimizer_config=dict(lr=5e-4)),
                dict(T=8, B=5, obs_spec={"obs": ((5,), "f32")}, action_dims=4, p_done=0.15, available_action=True)),
    # PopArt: the stored targets are de-normalised returns, normalised on entering the phase (:222-225)
    "auxpa": (dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=2, num_rnn_layers=0, popart=True, layernorm=True,
                   chunk_len=8, seed=83, value_dim=2),
              dict(popart=True, ppg_epochs=2, max_grad_norm=1.0, beta_clone=1.0, aux_value_head_weight=1.0),
              dict(T=16, B=4, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.1, value_dim=2)),
    # GRU backbones: chunked analysis from the stored states (:419-426)
    "auxgru": (dict(obs_dim=4, action_dim=3, hidden_dim=16, num_dense_layers=1, num_rnn_layers=1, popart=False, layernorm=True,
                    chunk_len=4, seed=84),
               dict(popart=False, ppg_epochs=2, max_grad_norm=10.0, beta_clone=1.0, aux_value_head_weight=1.0,
                    ppg_optimizer_config=dict(lr=1e-3)),
               dict(T=8, B=5, obs_spec=synthetic.CARTPOLE_OBS, action_dims=3, p_done=0.1,
                    policy_state={"actor_hx": (1, 16), "critic_hx": (1, 16)})),
}


def ppg_entry_arrays(arrays, T, seed, value_dim=1):
    """What MultiAgentPPG.step puts into its local cache (:193-203): the first T rows of obs / policy_state / info_mask / on_reset
    and a value target per row (here: drawn, the phase-1 code that would produce it cannot run)."""
    rng = np.random.default_rng(seed)
    e = {k: v[:T] for k, v in arrays.items() if k.startswith("obs.") or k.startswith("policy_state.") or k in ("info_mask", "on_reset")}
    e["value"] = (2.0 * rng.standard_normal((T, arrays["on_reset"].shape[1], value_dim))).astype(np.float32)
    # info_mask marks episode ends ("done" of the auxiliary loss, :264): make sure some rows are masked
    e["info_mask"] = (rng.random(e["info_mask"].shape) < 0.2).astype(np.uint8)
    return e


def gen_ppg():
    """The runnable pieces of the reference's Phasic Policy Gradient: the `actor-critic-auxiliary` policy's two analysis targets
    (actor_critic_policy.py:392-435) and the auxiliary phase -- `_compute_aux_loss`, backward, clip, `ppg_aux_optimizer.step()`
    (phasic_policy_gradient.py:205-243, 262-280), driven from here statement for statement because `MultiAgentPPG.step` itself
    raises before any arithmetic (:169 `_compute_adv` does not exist; `SampleBatch` drops :193's `value`): the cache entry is a
    plain NamedArray carrying the attributes `_compute_aux_loss` reads."""
    import legacy.algorithm.ppo.phasic_policy_gradient as ppg
    from oracle.ppg import OraclePPGAux
    out = {}
    for tag, (pargs, targs, skw) in PPG_CASES.items():
        trainer = api.trainer.make(api.config.Trainer("mappg", args=targs), api.config.Policy("actor-critic-auxiliary", args=pargs))
        if pargs.get("popart"):
            object.__setattr__(trainer.policy.net, "module", trainer.policy.net)
        net = trainer.policy.net
        sd0 = sd_to_np(net.state_dict())
        spec = check_init(dict(pargs, auxiliary_head=True, shared_backbone=False), net.state_dict())
        for k, v in sd0.items():
            out[f"{tag}_init_param:{k}"] = v
        T = skw["T"]
        arrays = synthetic.make_sample_arrays(seed=300, **skw)
        sample = ref_sample({k: v.copy() for k, v in arrays.items()})
        ts = recursive_apply(sample, lambda x: torch.from_numpy(x).float())
        # ---- analyze(target="ppg_ppo_phase") on the sample's first T rows
        with torch.no_grad():
            r1 = trainer.policy.analyze(ts[:T], target="ppg_ppo_phase")
        out[f"{tag}_p1_new_lp"], out[f"{tag}_p1_value"] = r1.new_action_log_probs.numpy(), r1.state_values.numpy()
        out[f"{tag}_p1_aux"], out[f"{tag}_p1_entropy"] = r1.aux_values.numpy(), r1.entropy.numpy()
        # ---- the cache entry and the auxiliary phase
        vd = pargs.get("value_dim", 1)
        e = ppg_entry_arrays(arrays, T, seed=7, value_dim=vd)
        if pargs.get("popart"):  # statistics away from their initial zeros
            with torch.no_grad():
                net.critic_head._PopArtValueHead__rms._RunningMeanStd__mean.copy_(torch.tensor([0.3, -0.2][:vd], dtype=torch.float64))
                net.critic_head._PopArtValueHead__rms._RunningMeanStd__mean_sq.copy_(torch.tensor([1.7, 2.5][:vd], dtype=torch.float64))
                net.critic_head._PopArtValueHead__rms._RunningMeanStd__debiasing_term.fill_(0.9)
            sd0 = sd_to_np(net.state_dict())
            for k, v in sd0.items():
                out[f"{tag}_init_param:{k}"] = v
        obs = NamedArray(**{k[4:]: v for k, v in e.items() if k.startswith("obs.")})
        ps = {k[len("policy_state."):]: v for k, v in e.items() if k.startswith("policy_state.")}
        entry_sample = NamedArray(obs=obs, policy_state=NamedArray(**ps) if ps else None, info_mask=e["info_mask"],
                                  on_reset=e["on_reset"], value=e["value"])
        entry = ppg._PPGLocalCache.CacheEntry(sample=entry_sample)
        to_t = lambda smp: recursive_apply(smp, lambda x: torch.from_numpy(x).to(dtype=torch.float32))
        # :209-225
        r2 = trainer.policy.analyze(to_t(entry.sample), target="ppg_aux_phase")
        entry.action_dists = [modules.distribution_detach_to_cpu(d) for d in r2.action_dists]
        if trainer.popart:
            entry.sample.value = trainer.policy.normalize_value(torch.from_numpy(entry.sample.value)).cpu().detach().numpy()
        for h, d in enumerate(r2.action_dists):
            out[f"{tag}_p2_logq{h}"] = d.logits.detach().numpy()
        out[f"{tag}_p2_aux"], out[f"{tag}_p2_pred"] = r2.auxiliary_value.detach().numpy(), r2.predicted_value.detach().numpy()
        out[f"{tag}_entry_value"], out[f"{tag}_entry_info_mask"] = e["value"], e["info_mask"]
        # the oracle on the same entry
        onet = OracleActorCritic(**pargs)
        onet.load_state_dict(sd0)
        oaux = OraclePPGAux(onet, beta_clone=targs.get("beta_clone", 1), aux_value_head_weight=targs.get("aux_value_head_weight", 1),
                            max_grad_norm=targs.get("max_grad_norm"), popart=targs.get("popart", False),
                            ppg_optimizer_config=targs.get("ppg_optimizer_config", {}))
        oaux.enter(e)
        for h, d in enumerate(r2.action_dists):
            assert torch.allclose(oaux.old[h], d.logits.detach(), rtol=1e-5, atol=1e-6), (tag, h)
        # The distributions were kept under the parameters above.  In the algorithm's own flow the first auxiliary epoch then runs
        # on the SAME parameters: KL = 0 and its gradient is float32 rounding noise, which Adam's first steps turn into full-size
        # moves -- nothing two implementations can agree on.  To pin the loss and its backward on general inputs the parameters are
        # moved first (a seeded perturbation, as if policy updates had happened in between), on both sides.
        prng = torch.Generator().manual_seed(1234)
        with torch.no_grad():
            for k, p_ in net.state_dict().items():
                if p_.dtype == torch.float32:
                    p_.add_(0.03 * torch.randn(p_.shape, generator=prng))
        sd1 = sd_to_np(net.state_dict())
        for k, v in sd1.items():
            out[f"{tag}_pert_param:{k}"] = v
        for k, p_ in onet.params.items():
            p_.data.copy_(torch.from_numpy(sd1[k]).to(p_.dtype))
        # :232-243
        version0 = trainer.policy.version
        names = ("auxiliary_value_loss", "value_head_loss", "policy_distance")
        for ep in range(targs["ppg_epochs"]):
            res = trainer.policy.analyze(to_t(entry.sample), target="ppg_aux_phase")
            aux_l, metrics = trainer._compute_aux_loss(entry, res)
            trainer.ppg_aux_optimizer.zero_grad()
            aux_l.backward()
            gn = None
            if trainer.max_grad_norm is not None:
                gn = torch.nn.utils.clip_grad_norm_(trainer.policy.parameters(), trainer.max_grad_norm)
            trainer.ppg_aux_optimizer.step()
            trainer.policy.inc_version()
            o = oaux.epoch(e)
            vals = [float(getattr(metrics, n).detach()) for n in names] + [float(aux_l.detach()), float(gn) if gn is not None else -1.0]
            for n, v in zip(names, vals):
                assert abs(o[n] - v) <= 1e-5 * max(1.0, abs(v)), (tag, ep, n, o[n], v)
            if gn is not None:
                assert abs(o["grad_norm"] - float(gn)) <= 2e-5 * max(1.0, float(gn)), (tag, ep, o["grad_norm"], float(gn))
            out[f"{tag}_epoch{ep}_terms"] = np.array(vals, dtype=np.float64)
        assert trainer.policy.version == version0 + targs["ppg_epochs"]
        sd = sd_to_np(net.state_dict())
        osd = onet.state_dict()
        for k, v in sd.items():
            assert np.allclose(osd[k].numpy(), v, rtol=1e-4, atol=1e-6), (tag, k)
            out[f"{tag}_final_param:{k}"] = v
        print(f"ppg case {tag}: terms {out[f'{tag}_epoch0_terms']}")
    out["term_names"] = np.array(["auxiliary_value_loss", "value_head_loss", "policy_distance", "loss", "grad_norm"])
    save("ppg.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["gae", "norm", "loss", "steps", "rollout", "host"]
    for w in which:
        globals()[f"gen_{w}"]()
