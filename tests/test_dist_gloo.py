"""World-size-2 checks of the data-parallel arithmetic on CPU (gloo), the way the reference tests its
distributed pieces (legacy/tests/modules_test.py:49-74,273-299): the statistic exchange that makes the
advantage normalisation global, the DDP gradient semantics (mean of per-rank masked means), and the
flat-buffer parameter broadcast.  The HIP kernels themselves need a GPU; here the collectives and the
host-side composition are exercised with the oracle's arithmetic standing in for the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ppo as oppo


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, adv, mask, expected, ckpt_flat):
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        half = adv.shape[1] // world
        a, m = adv[:, rank * half:(rank + 1) * half], mask[:, rank * half:(rank + 1) * half]
        # what MultiAgentPPO.step does with the three sums produced by srl_gae_scan: ONE all-reduce
        stats = torch.tensor(oppo.masked_stats(a, m), dtype=torch.float64)
        local_n = float(stats[0])
        g = stats.clone()
        dist.all_reduce(g)
        out = oppo.masked_normalization(a, m, stats=tuple(g.tolist()))
        np.testing.assert_almost_equal(out * m, expected[:, rank * half:(rank + 1) * half] * m, decimal=6)
        assert local_n == m.sum()
        # gradient semantics: each rank's gradient is of its LOCAL masked mean; all-reduce SUM then 1/world
        grad = torch.full((5,), float(rank + 1))
        dist.all_reduce(grad)
        assert torch.allclose(grad / world, torch.full((5,), (1 + 2) / 2))
        # parameter broadcast through the policy's own method on a CPU shell
        import srl_amd
        from srl_amd.api import config, policy as policy_api
        srl_amd.register_all()
        pol = policy_api.make(config.Policy("actor-critic", args=dict(obs_dim=4, action_dim=2, hidden_dim=8,
                                                                         num_dense_layers=1, num_rnn_layers=0,
                                                                         popart=False, seed=rank)))
        if rank == 0:
            pol.net.flat.copy_(ckpt_flat)
            pol._version = 11
        pol.distributed()
        assert torch.equal(pol.net.flat, ckpt_flat) and pol.version == 11
    finally:
        dist.destroy_process_group()


def test_two_rank_statistics_gradients_and_broadcast():
    rng = np.random.default_rng(0)
    adv = rng.standard_normal((6, 16, 1)).astype(np.float32)
    mask = (rng.random((6, 16, 1)) < 0.7).astype(np.float32)
    expected = oppo.masked_normalization(adv, mask)
    import srl_amd
    from srl_amd.algorithm.netspec import build_netspec
    spec, _ = build_netspec(obs_dim=4, action_dim=2, hidden_dim=8, num_dense_layers=1)
    flat = torch.arange(spec.total_params, dtype=torch.float32)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, adv, mask, expected, flat), nprocs=2, join=True)


def test_bucket_reducer_cuts_and_readiness():
    """Host logic of the overlapped gradient all-reduce: buckets tile the flat buffer in parameter order, a big tensor
    gets its own bucket, and a bucket fires exactly when the last of its parameters reports a final gradient."""
    from srl_amd.algorithm import netspec as ns
    from srl_amd.algorithm.mappo import _BucketReducer
    cnn = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
               popart=False, layernorm=False, shared_backbone=True,
               cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))
    spec, _ = ns.build_netspec(**cnn)

    class Net:
        pass

    net = Net()
    net.spec, net.grad = spec, torch.zeros(spec.total_params)
    r = _BucketReducer(net, 1 << 20)
    assert r.buckets[0][0] == 0 and r.buckets[-1][1] == spec.total_params
    assert all(r.buckets[i][1] == r.buckets[i + 1][0] for i in range(len(r.buckets) - 1))
    fc = spec.params["obs_modules_dict.obs.1._Convolution__model.7.0.weight"]
    assert [b[0] for b in r.buckets].count(fc.offset) == 1  # the 6.4 MB tensor starts a bucket of its own
    fired = []
    r._launch = lambda i: (fired.append(i), r.launched.__setitem__(i, True))
    cm = "obs_modules_dict.obs.1._Convolution__model"
    r.ready(["actor_head"]); r.ready(["critic_head"]); r.ready([f"{cm}.7.2"])
    assert fired == []  # the tail bucket also holds the FC bias
    r.ready([f"{cm}.7.0"])
    assert sorted(fired) == [1, 2]  # FC bias completes the tail bucket, FC weight its own: both go while the convs still run
    r.ready([f"{cm}.4"]); r.ready([f"{cm}.2"])
    assert sorted(fired) == [1, 2]
    r.ready([f"{cm}.0", "obs_modules_dict.obs.0"])
    # the bucket that becomes final LAST has nothing left to overlap with: `ready` holds it back for `finish`, which closes it in
    # stream order on the compute stream (round 6)
    assert sorted(fired) == [1, 2] and not r.launched[0] and not any(r.pending[0][0])
    r.LAST_INLINE = False   # the A/B switch: as before, launched from the last release
    r.begin()
    fired.clear()
    for pre in (["actor_head"], ["critic_head"], [f"{cm}.7.2"], [f"{cm}.7.0"], [f"{cm}.4"], [f"{cm}.2"], [f"{cm}.0", "obs_modules_dict.obs.0"]):
        r.ready(pre)
    assert sorted(fired) == [0, 1, 2] and fired[-1] == 0


def test_bucket_reducer_with_two_pipelines_waits_for_both():
    """Two row-chunk pipelines with their own gradient buffers: a bucket leaves when the last chunk of EACH pipeline has
    released it, whichever comes second, and each release leaves one stream marker for the fold to wait on."""
    from srl_amd.algorithm import netspec as ns
    from srl_amd.algorithm.mappo import _BucketReducer
    spec, _ = ns.build_netspec(obs_dim=4, action_dim=3, hidden_dim=64, num_dense_layers=2, shared_backbone=False)

    class Net:
        pass

    net = Net()
    net.spec, net.grad = spec, torch.zeros(spec.total_params)
    r = _BucketReducer(net, 8192)
    assert len(r.buckets) >= 3
    fired, marks = [], []
    r._launch = lambda i: (fired.append(i), r.launched.__setitem__(i, True))
    r._mark = lambda: marks.append(len(marks)) or len(marks)
    r.begin([net.grad, torch.zeros(spec.total_params)])
    every = sorted({p for b in r.buckets for p in b[2]})
    last = r.buckets[-1][2]
    both = [i for i, b in enumerate(r.buckets) if b[2] <= last]  # (a bucket may hold only the bias of the tail's first layer)
    r.hook(0)(sorted(last))  # pipeline 0 releases the tail bucket: pipeline 1 still holds it
    assert fired == [] and len(marks) == len(both) < len(r.buckets) and all(r.events[0][i] is not None for i in both)
    r.hook(1)(every)  # pipeline 1 releases everything: only what both have released goes
    assert fired == both and len(marks) == len(both) + len(r.buckets)
    r.hook(0)(every)
    held = [i for i in range(len(r.buckets)) if not r.launched[i]]   # the one completed last waits for `finish` (stream order)
    assert len(held) == 1 and sorted(fired + held) == list(range(len(r.buckets))) and len(marks) == 2 * len(r.buckets)
    assert all(ev is not None for pipe in r.events for ev in pipe)
    r.begin()  # one pipeline again: no markers
    r.ready(every)
    assert len(marks) == 2 * len(r.buckets) and sum(r.launched) == len(r.buckets) - 1
