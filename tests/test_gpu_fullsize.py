"""BASELINE.json configs at FULL size -- [1] 512 envs x 128 steps and [2] 4096 envs x 128 steps (uint8 (4,84,84) frames,
NatureCNN-512), [3] SMAC 3m 1024 x 3 agents x 100 steps, [4] football 256 envs (per GPU) x 200 steps CNN + LSTM: the
oracle's network cannot run these in seconds, so parity is checked through size-independent properties of the path
(chunking invariance, repeatability, linearity) plus the numpy oracle where it is cheap (GAE returns, statistics)."""
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


def device_sample(seed, b=B, frame=(4, 84, 84), t=T, actions=6, p_done=1.0 / 800, **kw):
    arr = synthetic.make_sample_arrays(seed=seed, T=t, B=b, obs_spec={}, action_dims=actions, p_done=p_done, **kw)
    dev = {k: torch.from_numpy(v).cuda() for k, v in arr.items()}
    gen = torch.Generator(device="cuda").manual_seed(seed)
    dev["obs.obs"] = torch.randint(0, 256, (t + 1, b, *frame), dtype=torch.uint8, device="cuda", generator=gen)
    return dev, arr


def check_gae_vs_oracle(sample, arr, gamma=0.99, lmbda=0.97, popart_net=None):
    """The advantages / value targets the step wrote back into the sample against the numpy oracle (1e-5 relative)."""
    from oracle import gae as ogae
    value = arr["analyzed_result.value"]
    if popart_net is not None:  # PopArt: the trace runs on de-normalised values (mappo.py:120-124; popart.py:53-59)
        rms = popart_net.popart_state.cpu().numpy()
        debias = max(rms[2], 1e-5)
        mean, var = rms[0] / debias, max(rms[1] / debias - (rms[0] / debias)**2, 1e-2)
        value = (value.astype(np.float64) * np.sqrt(var) + mean).astype(np.float32)
    o_adv, o_ret = ogae.adv_and_value_target(arr["reward"], value, arr["truncated"], arr["done"], arr["on_reset"], gamma, lmbda)
    adv, ret = sample.analyzed_result.adv.cpu().numpy(), sample.analyzed_result.ret.cpu().numpy()
    Tn = o_adv.shape[0]
    assert (np.abs(adv[:Tn] - o_adv) <= 1e-5 * np.maximum(np.abs(o_adv), 1.0)).all()
    assert (np.abs(ret[:Tn] - o_ret) <= 1e-5 * np.maximum(np.abs(o_ret), 1.0)).all()
    assert (adv[Tn:] == 0).all() and (ret[Tn:] == 0).all()  # the zero pad row (mappo.py:254-256)


def check_analyze_rows_vs_oracle(tr, smp, sample, arr, picks=64):
    """Full-size NETWORK outputs against the oracle, not only properties: `policy.analyze(target="ppo")` over ALL 4096 x 128 rows
    of the sample (actor_critic_policy.py:338-390; the encoder walks them in pieces of 32 768 rows on the kernels the benchmark
    times), then new log-probability, state value and entropy of 64 sampled (t, b) positions against the float32
    `OracleActorCritic` evaluated on just those rows, 1e-5 relative (BASELINE.json's bar)."""
    from oracle.net import OracleActorCritic
    t_all, b_all = arr["reward"].shape[0] - 1, arr["reward"].shape[1]
    ar = tr.policy.analyze(smp[:t_all], target="ppo")
    rng = np.random.default_rng(123)
    # spread over the pieces and row chunks: the first and last rows of the sample, the rest uniformly
    ts = np.concatenate([[0, t_all - 1], rng.integers(0, t_all, picks - 2)])
    bs = np.concatenate([[0, b_all - 1], rng.integers(0, b_all, picks - 2)])
    got = [x[ts, bs].detach().cpu().numpy().reshape(picks) for x in (ar.new_action_log_probs, ar.state_values, ar.entropy)]
    onet = OracleActorCritic(**POLICY)
    onet.load_state_dict({k: v.numpy() for k, v in tr.policy.get_checkpoint()["state_dict"].items()})
    frames = sample["obs.obs"][torch.from_numpy(ts).cuda(), torch.from_numpy(bs).cuda()].cpu()   # [picks, 4, 84, 84] uint8
    action = torch.from_numpy(arr["action.x"][ts, bs].astype(np.int64))[None]
    with torch.no_grad():
        lp, val, ent, _ = onet.analyze({"obs": frames[None].float()}, action, None)
    for name, g, o in zip(("log-prob", "value", "entropy"), got, (lp, val, ent)):
        o = o.numpy().reshape(picks)
        assert (np.abs(g - o) <= 1e-5 * np.maximum(np.abs(o), 1.0)).all(), (name, float(np.abs(g - o).max()))
    # the analysis taped every piece of the whole sample (~50 GB of activations): hand it back before the step allocates its own
    del ar
    tr.policy._analysis = None
    tr.policy.net._tape = None
    tr.policy.net.ws._bufs.clear()
    torch.cuda.empty_cache()


def check_recurrent_chunks_vs_oracle(tr, smp, sample, arr, policy_args, picks=7, tol=1e-5):
    """Full-size outputs of a RECURRENT net against the oracle: `policy.analyze(target="ppo")` over the whole sample, then sampled
    (chunk, column) pairs -- a chunk of `chunk_len` steps starts from the state stored with its first row
    (actor_critic_policy.py:349-380), so it is an independent little sample of its own -- through the float32
    `OracleActorCritic`: new log-probability, state value and entropy of picks x chunk_len rows at 1e-5 relative."""
    from oracle.net import OracleActorCritic
    t_all, b_all = arr["reward"].shape[0] - 1, arr["reward"].shape[1]
    C = policy_args.get("chunk_len", 10)
    ar = tr.policy.analyze(smp[:t_all], target="ppo")
    rng = np.random.default_rng(321)
    cs = np.concatenate([[0, t_all // C - 1], rng.integers(0, t_all // C, picks - 2)])
    bs = np.concatenate([[0, b_all - 1], rng.integers(0, b_all, picks - 2)])
    rows = np.stack([np.arange(c * C, (c + 1) * C) for c in cs], 1)           # [C, picks]
    got = [x.detach().cpu().numpy()[rows, bs[None, :]].reshape(C, picks) for x in (ar.new_action_log_probs, ar.state_values, ar.entropy)]
    onet = OracleActorCritic(**policy_args)
    onet.load_state_dict({k: v.numpy() for k, v in tr.policy.get_checkpoint()["state_dict"].items()})
    tr_, tb_ = torch.from_numpy(rows).cuda(), torch.from_numpy(bs).cuda()[None, :].expand(C, picks)
    frames = sample["obs.obs"][tr_, tb_].cpu().float()                      # [C, picks, ...]
    action = torch.from_numpy(arr["action.x"][rows, bs[None, :]].astype(np.int64))
    on_reset = torch.from_numpy(arr["on_reset"][rows, bs[None, :]].astype(np.float32))
    state = tuple(torch.from_numpy(arr[f"policy_state.{k}"][rows, bs[None, :]]) for k in ("actor_hx", "critic_hx"))
    with torch.no_grad():
        lp, val, ent, _ = onet.analyze({"obs": frames}, action, on_reset, state)
    if tr.policy.net.spec.popart:   # analyze returns the head's raw (normalised) values on both sides
        pass
    for name, g, o in zip(("log-prob", "value", "entropy"), got, (lp, val, ent)):
        o = o.numpy().reshape(C, picks)
        assert (np.abs(g - o) <= tol * np.maximum(np.abs(o), 1.0)).all(), (name, float(np.abs(g - o).max()))
    del ar
    tr.policy._analysis = None
    tr.policy.net._tape = None


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
    sample, _ = device_sample(7)
    results = []
    from conftest import BENCH_CHUNK_TILES
    for chunk in (16384, 8192, 16384):
        tr = make(chunk)
        hip.dispatch_tiles(reset=True)
        res = tr.step(synthetic.to_sample_batch(dict(sample)))
        nchunks = -(-T * B // chunk)
        assert hip.dispatch_tiles(reset=True) == {k: v * nchunks for k, v in BENCH_CHUNK_TILES.items()}, chunk
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


def test_config2_step_4096_envs():
    """BASELINE.json configs[2], the configuration the metric is quoted on: ONE GPU takes all 4096 env columns x 128 steps
    (14.9 GB of uint8 frames, 32 row-chunks).  GAE returns against the numpy oracle, row-chunking invariance (16384 vs
    32768 rows per chunk), repeatability, and the statistics the step reports."""
    b = 4096
    sample, arr = device_sample(11, b=b)
    results = []
    for chunk in (16384, 32768, 16384):
        tr = make(chunk)
        smp = synthetic.to_sample_batch(dict(sample))
        if not results:
            check_analyze_rows_vs_oracle(tr, smp, sample, arr)
        res = tr.step(smp)
        results.append((res.stats, tr.policy.get_checkpoint()["state_dict"]))
        if chunk == 16384 and len(results) == 1:
            check_gae_vs_oracle(smp, arr)
        del tr, smp, res
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    (s0, p0), (s1, p1), (s2, p2) = results
    for k in s0:
        tol = 1e-5 if k in ("policy_loss", "value_loss", "entropy") else 1e-4
        assert abs(s0[k] - s1[k]) <= tol * max(abs(s0[k]), 1e-3), ("chunking", k, s0[k], s1[k])
        assert abs(s0[k] - s2[k]) <= 1e-6 * max(abs(s0[k]), 1e-3), ("repeat", k, s0[k], s2[k])
    for k in p0:
        assert torch.allclose(p0[k], p1[k], rtol=0, atol=2e-5), ("chunking", k)
        assert torch.allclose(p0[k], p2[k], rtol=0, atol=1e-6), ("repeat", k)
    assert s0["frames"] == T * b and np.isfinite(list(s0.values())).all()
    mask = 1.0 - arr["on_reset"][1:].astype(np.float64)
    assert abs(s0["done"] - float(arr["done"][:T].mean())) < 1e-9 and abs(s0["truncated"] - float(arr["truncated"][:T].mean())) < 1e-9
    assert mask.sum() > 0.99 * T * b
    del sample
    torch.cuda.empty_cache()


def test_config4_football_per_gpu_size():
    """BASELINE.json configs[4] at its per-GPU size: 256 of the 2048 football environments x 200 steps, the
    `football-smm-separate` preset with an LSTM ((4, 96, 72) uint8 frames -> convolution stack -> the halving Linear
    tower -> LSTM-128, separate actor / critic, PopArt; 676 M parameters).  GAE returns against the numpy oracle (on the
    values de-normalised with the initial PopArt statistics, restated in numpy), invariance of the update to how
    the encoder rows are cut into pieces, repeatability.  The orthogonal initialisation (a QR of a 22528 x 11264 matrix,
    minutes on one core) is replaced by a scaled normal draw: random-init weights of the same architecture."""
    import math
    Tf, Bf, H = 200, 256, 128
    orig = torch.nn.init.orthogonal_

    def cheap(t, gain=1.0):
        with torch.no_grad():
            return t.normal_(0.0, gain / math.sqrt(t.shape[1] if t.dim() > 1 else t.numel()))

    torch.nn.init.orthogonal_ = cheap
    try:
        tr = trainer_api.make(config.Trainer("mappo", args=dict(popart=True, clip_value=True, value_loss="huber",
                                                                value_loss_config=dict(delta=10.0), max_grad_norm=10.0,
                                                                optimizer_config=dict(lr=5e-4, eps=1e-5))),
                              config.Policy("football-smm-separate", args=dict(rnn_type="lstm", seed=1)))
    finally:
        torch.nn.init.orthogonal_ = orig
    net = tr.policy.net
    assert net.spec.total_params > 600e6
    sample, arr = device_sample(5, b=Bf, frame=(4, 96, 72), t=Tf, actions=19, p_done=1 / 400,
                                policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
    flat0, pop0 = net.flat.clone(), net.popart_state.clone()

    def one_step(rows):
        net.flat.copy_(flat0)
        net.popart_state.copy_(pop0)
        tr._m.zero_(), tr._v.zero_()
        tr._opt_steps = 0
        tr.policy._popart_updates = 0
        net.encoder_rows = rows
        smp = synthetic.to_sample_batch(dict(sample))
        res = tr.step(smp)
        return res.stats, net.flat.clone(), smp

    rows = net.encoder_rows
    pop_init = type("P", (), {"popart_state": pop0})
    # the network's outputs at full size against the oracle (round 6: the dense tower runs on the pre-split kernels,
    # `HipNet._linear_fwd_h2d`): 7 sampled chunks of 10 steps, each from its stored LSTM state
    from srl_amd.algorithm.game_policies import FootballSMMPolicy
    torch.nn.init.orthogonal_ = cheap   # (the oracle's constructor initialises 676 M parameters too, before the checkpoint replaces them)
    try:
        # (4e-5: two float32 evaluations of an eight-layer 22 528-wide tower + six LayerNorms + an LSTM; the layer-by-layer
        # kernels sit at 2.2e-5 of the same oracle, the pre-split ones at 2.6e-5 -- `scripts/football_grad_check.py` holds both
        # against the float64 oracle, gradients included)
        check_recurrent_chunks_vs_oracle(tr, synthetic.to_sample_batch(dict(sample)), sample, arr,
                                         dict(FootballSMMPolicy.defaults, rnn_type="lstm", seed=1,
                                              cnn_layers=dict(obs=[(4, 5, 1, 0, "zeros"), (8, 3, 1, 0, "zeros"), (4, 3, 1, 0, "zeros")])),
                                         tol=4e-5)
    finally:
        torch.nn.init.orthogonal_ = orig
    s0, p0, smp = one_step(rows)
    check_gae_vs_oracle(smp, arr, popart_net=pop_init)
    s1, p1, _ = one_step(rows // 2 + 17)  # more, unevenly cut encoder pieces
    s2, p2, _ = one_step(rows)
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "importance_weight", "denorm_value"):
        assert abs(s0[k] - s1[k]) <= 1e-4 * max(abs(s0[k]), 1e-3), ("pieces", k, s0[k], s1[k])
        assert abs(s0[k] - s2[k]) <= 1e-6 * max(abs(s0[k]), 1e-3), ("repeat", k, s0[k], s2[k])
    assert s0["frames"] - 0 == Tf * Bf * 1 or s0["frames"] % (Tf * Bf) == 0
    d1, d2 = (p0 - p1).abs(), (p0 - p2).abs()
    assert float(d1.max()) <= 5e-4 and float((d1 > 2e-5).float().mean()) < 1e-3, float(d1.max())
    assert float(d2.max()) <= 1e-6
    assert np.isfinite(list(s0.values())).all() and float((p0 - flat0).abs().max()) > 0  # the step moved the weights
    del tr, net, sample
    torch.cuda.empty_cache()


def test_smac_config_full_size_step_vs_oracle():
    """BASELINE.json configs[3] at full size (1024 shared 3m environments x 3 agents x 100 steps, `smac_rnn`): the GAE
    returns of all 307 200 agent-steps against the numpy oracle (1e-5), the folded-agent step equal to the same data
    presented as 3072 independent columns, and the CPU oracle's full step (LSTM unrolled on the CPU, seconds at this
    size) on the loss terms."""
    from oracle.net import OracleSMACNet
    from oracle.trainer import OracleMappo
    Ts, Bs, A, H = 100, 1024, 3, 64
    pol = dict(map_name="3m", hidden_dim=H, chunk_len=10, seed=1, shared=True)
    tr_args = dict(popart=True, clip_value=True, dual_clip=False, value_loss="huber", value_loss_config=dict(delta=10.0),
                   max_grad_norm=10.0, optimizer_config=dict(lr=5e-4, eps=1e-5))
    arrays = synthetic.make_multiagent_arrays(seed=4, T=Ts, B=Bs, agents=A,
                                              obs_spec={"local_obs": ((30,), "f32"), "state": ((48,), "f32")}, action_dim=9,
                                              p_done=1 / 60, policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
    shared = trainer_api.make(config.Trainer("mappo", args=tr_args), config.Policy("smac_rnn", args=pol))
    onet = OracleSMACNet(30, 48, 9, H, 10)
    onet.load_state_dict({k: v.numpy() for k, v in shared.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **tr_args)
    sample = synthetic.to_sample_batch({k: v.copy() for k, v in arrays.items()})
    res = shared.step(sample)
    ostats, oout = oracle.step(arrays)
    assert sample.analyzed_result.ret.shape == (Ts + 1, Bs, A, 1)
    err = np.abs(sample.analyzed_result.ret - oout["ret"]) / np.maximum(np.abs(oout["ret"]), 1.0)
    assert err.max() <= 1e-5, err.max()
    assert res.stats["frames"] == Ts * Bs * A
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "importance_weight", "denorm_value"):
        assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (k, res.stats[k], ostats[k])
    # the same rows as 3072 independent columns through the non-shared policy (is_alive is not part of that observation,
    # so compare without it on both sides)
    flat = {k: v.reshape(v.shape[0], Bs * A, *v.shape[3:]) for k, v in arrays.items() if k != "obs.is_alive"}
    folded = {k: v for k, v in arrays.items() if k != "obs.is_alive"}
    t1 = trainer_api.make(config.Trainer("mappo", args=tr_args), config.Policy("smac_rnn", args=dict(pol, shared=False)))
    t2 = trainer_api.make(config.Trainer("mappo", args=tr_args), config.Policy("smac_rnn", args=pol))
    r1, r2 = t1.step(synthetic.to_sample_batch(flat)), t2.step(synthetic.to_sample_batch(folded))
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm"):
        assert abs(r1.stats[k] - r2.stats[k]) <= 1e-6 * max(1.0, abs(r2.stats[k])), k
    d = (t1.policy.net.flat - t2.policy.net.flat).abs()  # split-K partial sums land in launch order: not bitwise
    assert float(d.max()) <= 5e-4 and float((d > 1e-6).float().mean()) < 1e-3


def test_config0_towers_at_the_metric_batch_vs_oracle():
    """BASELINE.json configs[0]'s nets (separate 2 x 64 actor / critic towers on 4 observations) at the METRIC's batch -- 4096 envs x
    128 steps, the update `roofline_mlp` times: each tower one launch per direction on the kernels instantiated for its shape
    (csrc/mlp_sig.h), no activation tape.  The whole step against the CPU oracle on all 524 288 rows: GAE returns 1e-5, loss terms
    and gradient norm 2e-5, and the parameters after the step (Adam's first step moves an element by ~lr whatever its gradient's
    size: elements whose gradient is zero to rounding aside, they agree)."""
    from oracle.net import OracleActorCritic
    from oracle.trainer import OracleMappo
    from srl_amd import hip
    T, B = 128, 4096
    pol = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False, layernorm=False,
               shared_backbone=False, seed=1)
    tr_args = dict(popart=False, optimizer_config=dict(lr=3e-4))
    tr = trainer_api.make(config.Trainer("mappo", args=tr_args), config.Policy("actor-critic-separate", args=pol))
    onet = OracleActorCritic(**{k: v for k, v in pol.items() if k not in ("seed", "popart")})
    onet.load_state_dict({k: v.numpy() for k, v in tr.policy.get_checkpoint()["state_dict"].items()})
    oracle = OracleMappo(onet, **tr_args)
    arrays = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05)
    sample = synthetic.to_sample_batch({k: torch.from_numpy(v).cuda() for k, v in arrays.items()})
    hip.dispatch_counts(reset=True)
    res = tr.step(sample)
    ostats, oout = oracle.step(arrays)
    ret = sample.analyzed_result.ret.cpu().numpy()
    err = np.abs(ret - oout["ret"]) / np.maximum(np.abs(oout["ret"]), 1.0)
    assert err.max() <= 1e-5, err.max()
    for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "importance_weight"):
        assert abs(res.stats[k] - ostats[k]) <= 2e-5 * max(abs(ostats[k]), 1e-2), (k, res.stats[k], ostats[k])
    got = tr.policy.get_checkpoint()["state_dict"]
    far = total = 0
    for k, v in onet.state_dict().items():
        d = np.abs(got[k].cpu().numpy().astype(np.float64) - (v.detach().numpy() if hasattr(v, 'detach') else np.asarray(v)).astype(np.float64))
        far += int((d > 1e-6).sum())
        total += d.size
        assert d.max() <= 2 * 3e-4 * 1.01, (k, float(d.max()))
    assert far <= max(2, total // 100), (far, total)
