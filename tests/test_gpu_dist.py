"""The data-parallel trainer path on device memory: two processes share the one GPU of the test box, each with its own
columns of the batch, joined by a gloo process group (RCCL needs one device per rank; the collectives the trainer issues
are the same calls either way: the advantage / PopArt statistics all-reduces, the bucketed gradient all-reduce
overlapped with backward, the parameter broadcast).  Checked against the CPU oracle's emulation of the reference's
DistributedDataParallel semantics (``OracleMappo.step_dp``)."""
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

POLICY = dict(obs_dim=4, action_dim=3, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=True, layernorm=True,
              shared_backbone=False, chunk_len=8)
TRAINER = dict(popart=True, ppo_epochs=2, clip_value=True, dual_clip=False, value_loss="huber",
               value_loss_config=dict(delta=10.0), optimizer_config=dict(lr=1e-3), max_grad_norm=5.0,
               grad_bucket_bytes=8192, chunk_rows=100)
T, B, STEPS = 24, 12, 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_arrays(step, rank, world, unequal=False):
    from srl_amd.runtime import synthetic
    if unequal:  # rank 1's environments reset ~15x as often: the ranks' loss masks hold very different counts
        return synthetic.make_sample_arrays(seed=90 + 10 * step + rank, T=T, B=B // world, obs_spec=synthetic.CARTPOLE_OBS,
                                            action_dims=3, p_done=0.02 if rank == 0 else 0.3)
    full = synthetic.make_sample_arrays(seed=50 + step, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=3, p_done=0.1)
    half = B // world
    return {k: np.ascontiguousarray(v[:, rank * half:(rank + 1) * half]) for k, v in full.items()}


def _worker(rank, world, port, out, unequal=False, pipelines=2, chunk_rows=100):
    import srl_amd
    from srl_amd.api import config, trainer as trainer_api
    from srl_amd.runtime import synthetic
    srl_amd.register_all()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        # different seeds: joining the group must adopt rank 0's parameters (what the DDP constructor does)
        trainer = trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, pipelines=pipelines, chunk_rows=chunk_rows)),
                                   config.Policy("actor-critic", args=dict(POLICY, seed=7 + rank)))
        trainer.distributed(rank=rank, world_size=world, init_method=None)
        init = {k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()}
        stats = []
        for step in range(STEPS):
            res = trainer.step(synthetic.to_sample_batch(_rank_arrays(step, rank, world, unequal)))
            stats.append(res.stats)
        final = {k: v.numpy() for k, v in trainer.policy.get_checkpoint()["state_dict"].items()}
        out[rank] = dict(init=init, final=final, stats=stats, reducer=dict(trainer._reducer.stats),
                         buckets=len(trainer._reducer.buckets), pipes=1 + len(trainer._twin or []))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("unequal,pipelines,chunk_rows", [(False, 2, 100), (True, 2, 100), (False, 1, 100), (False, 4, 30)],
                         ids=["split-batch", "unequal-mask-counts", "one-pipeline", "default-four-pipelines"])
def test_two_rank_trainer_matches_ddp_semantics(unequal, pipelines, chunk_rows):
    """144 rows per rank in chunks of 100: with the default two pipelines each chunk is the LAST chunk of its pipeline, so
    both backward passes release buckets (`_BucketReducer.ready` on device, one event per pipeline and bucket) and every
    bucket's all-reduce is launched from inside the second backward pass after the other pipeline's slice was folded in;
    with one pipeline the second chunk's backward releases them.  `unequal`: the ranks' masks hold very different counts.  Reference semantics (mappo.py:184,197,199 under DDP): each
    rank divides its masked sums by its LOCAL count, the gradients are then averaged over ranks with equal weight, while the
    advantage normalisation uses the GLOBAL statistics -- not a global masked mean of the loss.
    `default-four-pipelines`: 144 rows per rank in chunks of 30 = five chunks on the trainer's DEFAULT four pipelines (the first
    executor takes chunks 0 and 4, its three twins one each): a bucket leaves when the fourth pipeline has released it, after
    THREE twin slices were folded into the first buffer on the reduction stream -- on device, nothing left for `finish`."""
    from oracle.net import OracleActorCritic
    from oracle.trainer import OracleMappo
    world = 2
    if unequal:
        counts = [int((1 - _rank_arrays(0, r, world, True)["on_reset"][1:]).sum()) for r in range(world)]
        assert counts[0] > 1.15 * counts[1], counts
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out, unequal, pipelines, chunk_rows), nprocs=world, join=True)
        res = {r: out[r] for r in range(world)}
    for r in range(world):  # the overlap ran: every bucket of every epoch left from inside a backward pass
        red, nb = res[r]["reducer"], res[r]["buckets"]
        epochs = STEPS * TRAINER["ppo_epochs"]
        # (every bucket but the one that becomes final last: that one has nothing left to overlap with and is closed in stream
        # order behind the chunk loop, `_BucketReducer.finish`)
        assert nb >= 3 and red["launched_in_backward"] == (nb - 1) * epochs and red["launched_in_finish"] == 0, red
        assert red["closed_inline"] == epochs, red
        assert res[r]["pipes"] == pipelines, res[r]["pipes"]
        assert red["slices_folded"] == nb * epochs * (pipelines - 1), red
    for k in res[0]["init"]:  # both ranks start from rank 0's parameters and stay identical
        assert np.array_equal(res[0]["init"][k], res[1]["init"][k]), k
        assert np.array_equal(res[0]["final"][k], res[1]["final"][k]), k
    onet = OracleActorCritic(**POLICY)
    onet.load_state_dict(res[0]["init"])
    oracle = OracleMappo(onet, **{k: v for k, v in TRAINER.items() if k not in ("grad_bucket_bytes", "chunk_rows")})
    for step in range(STEPS):
        ostats, _ = oracle.step_dp([_rank_arrays(step, r, world, unequal) for r in range(world)])
        for r in range(world):
            got = res[r]["stats"][step]
            for k in ("policy_loss", "value_loss", "entropy", "grad_norm", "importance_weight", "clip_ratio", "denorm_value"):
                assert abs(got[k] - ostats[r][k]) <= 2e-5 * max(abs(ostats[r][k]), 1e-2), (step, r, k, got[k], ostats[r][k])
    osd = onet.state_dict()
    for k, v in res[0]["final"].items():
        assert np.abs(v - osd[k].numpy()).max() <= 3e-5, (k, np.abs(v - osd[k].numpy()).max())


PPG_POLICY = dict(obs_dim=4, action_dim=[3, 2], hidden_dim=32, num_dense_layers=1, num_rnn_layers=0, popart=False, layernorm=True,
                  chunk_len=8, seed=81)
PPG_TRAINER = dict(popart=False, ppg_epochs=3, max_grad_norm=5.0, beta_clone=1.0, aux_value_head_weight=0.5,
                   ppg_optimizer_config=dict(lr=1e-3), grad_bucket_bytes=2048, chunk_rows=40)


def _ppg_entry(rank):
    from srl_amd.runtime import synthetic
    T, Bq = 16, 6
    arr = synthetic.make_sample_arrays(seed=400 + rank, T=T, B=Bq, obs_spec=synthetic.CARTPOLE_OBS, action_dims=[3, 2],
                                       p_done=0.1 if rank == 0 else 0.3)
    rng = np.random.default_rng(500 + rank)
    e = {k: v[:T] for k, v in arr.items() if k.startswith("obs.") or k == "on_reset"}
    e["value"] = rng.standard_normal((T, Bq, 1)).astype(np.float32)
    e["info_mask"] = (rng.random((T, Bq, 1)) < 0.2).astype(np.uint8)
    return e


def _ppg_perturbation(state):
    rng = np.random.default_rng(77)
    return {k: (np.asarray(v) + 0.05 * rng.standard_normal(np.asarray(v).shape).astype(np.float32)) for k, v in state.items()}


def _ppg_worker(rank, world, port, out):
    import srl_amd
    from srl_amd.algorithm.mappg import _CacheEntry
    from srl_amd.api import config, trainer as trainer_api
    srl_amd.register_all()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        trainer = trainer_api.make(config.Trainer("mappg", args=PPG_TRAINER),
                                   config.Policy("actor-critic-auxiliary", args=dict(PPG_POLICY, seed=81 + rank)))
        trainer.distributed(rank=rank, world_size=world, init_method=None)
        pol = trainer.policy
        init = {k: v.numpy() for k, v in pol.get_checkpoint()["state_dict"].items()}
        e = _ppg_entry(rank)
        entry = _CacheEntry({k[4:]: v for k, v in e.items() if k.startswith("obs.")}, None, e["info_mask"], e["on_reset"], e["value"])
        trainer.enter_aux_phase(entry)
        # (the first epoch would otherwise start at KL = 0: rounding-noise gradients that Adam turns into full-size steps)
        pol.load_checkpoint(dict(steps=pol.version, state_dict={k: torch.from_numpy(v) for k, v in _ppg_perturbation(init).items()}))
        terms = []
        for _ in range(PPG_TRAINER["ppg_epochs"]):
            m = trainer.aux_epoch(entry)
            terms.append((m.auxiliary_value_loss, m.value_head_loss, m.policy_distance, trainer.last_aux_grad_norm))
        final = {k: v.numpy() for k, v in pol.get_checkpoint()["state_dict"].items()}
        out[rank] = dict(init=init, final=final, terms=terms, reducer=dict(trainer._reducer.stats), buckets=len(trainer._reducer.buckets))
    finally:
        dist.destroy_process_group()


def test_two_rank_ppg_auxiliary_phase_matches_ddp_semantics():
    """`mappg` under data parallelism (phasic_policy_gradient.py:108: MultiAgentPPG inherits MultiAgentPPO.distributed; the policy is
    DDP in the auxiliary phase too): two ranks, each with the cache entry of ITS sample (different mask counts), three auxiliary
    epochs in chunks of 40 rows.  Against `OraclePPGAux.epoch_dp`: every rank's loss terms are masked means over its LOCAL rows, the
    gradients are averaged over ranks before the clip and the auxiliary optimiser's step; both ranks end with identical
    parameters, the oracle's to 2e-5; every bucket left from inside the backward pass of the last chunk."""
    from oracle.net import OracleActorCritic
    from oracle.ppg import OraclePPGAux
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_ppg_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = {r: out[r] for r in range(world)}
    for k in res[0]["init"]:
        assert np.array_equal(res[0]["init"][k], res[1]["init"][k]), k       # rank 0's parameters everywhere
        assert np.array_equal(res[0]["final"][k], res[1]["final"][k]), k
    for r in range(world):
        red, nb = res[r]["reducer"], res[r]["buckets"]
        assert nb >= 2 and red["launched_in_backward"] == (nb - 1) * PPG_TRAINER["ppg_epochs"] and red["launched_in_finish"] == 0, red
        assert red["closed_inline"] == PPG_TRAINER["ppg_epochs"], red
    onet = OracleActorCritic(**PPG_POLICY, auxiliary_head=True)
    onet.load_state_dict(res[0]["init"])
    oracle = OraclePPGAux(onet, beta_clone=1.0, aux_value_head_weight=0.5, max_grad_norm=5.0, popart=False,
                          ppg_optimizer_config=dict(lr=1e-3))
    entries = [_ppg_entry(r) for r in range(world)]
    oracle.enter_dp(entries)
    for k, v in _ppg_perturbation(res[0]["init"]).items():   # in place: the oracle's optimiser holds these tensors
        onet.params[k].data.copy_(torch.from_numpy(v).to(onet.params[k].dtype))
    for ep in range(PPG_TRAINER["ppg_epochs"]):
        outs, gn = oracle.epoch_dp(entries)
        for r in range(world):
            got = res[r]["terms"][ep]
            for i, k in enumerate(("auxiliary_value_loss", "value_head_loss", "policy_distance")):
                assert abs(got[i] - outs[r][k]) <= 1e-5 * max(abs(outs[r][k]), 1e-2), (ep, r, k, got[i], outs[r][k])
            assert abs(got[3] - gn) <= 2e-5 * max(gn, 1e-2), (ep, r, got[3], gn)
    osd = onet.state_dict()
    for k, v in res[0]["final"].items():
        assert np.abs(v - osd[k].numpy()).max() <= 2e-5, (k, np.abs(v - osd[k].numpy()).max())


def test_bench_script_with_two_ranks():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), except that
    the two ranks share this box's GPU over gloo: every rank must reach the same collectives in the same order (warm-up,
    timed steps, the untimed profiled step, the final barrier) and rank 0 must print one JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRL_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--global-envs", "32", "--rollout-len", "8", "--chunk-rows", "64"]  # 128 rows per rank: one chunk per pipeline
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    assert line["config"]["global_envs"] == 32 and line["config"]["envs_per_gpu"] == 16 and line["value"] > 0
    assert line["config"]["collective_ranks"] == 2 and line["from_pinned_host"]["value"] > 0
    assert line["resident_in_hbm"]["value"] > 0 and line["ring_fed"]["obs_ring"]["rows_patched"] == 0  # every stamp alive
    assert line["roofline"]["bound"] == "mfma" and "cpu_baseline" not in line  # the CPU baseline is N = 1 only
    # the NatureCNN's buckets leave from inside the backward passes of the two pipelines' last chunks, layer by layer
    gb = line["config"]["grad_buckets"]
    assert line["config"]["pipelines"] >= 2 and gb["buckets"] >= 2 and gb["launched_in_finish"] == 0, gb
    assert gb["launched_in_backward"] == (gb["buckets"] - 1) * gb["epochs"] and gb["closed_inline"] == gb["epochs"], gb
    assert 0.8 * gb["buckets"] * gb["epochs"] <= gb["slices_folded"] <= gb["buckets"] * gb["epochs"], gb  # (one-chunk legs: no fold)


def test_bench_script_one_rank_over_rccl():
    """The same script with a real RCCL process group (one rank: all this box can offer): communicator creation, the
    float64 statistics all-reduces, the asynchronous bucket all-reduces on gradient slices, the barriers."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--global-envs", "16", "--rollout-len", "8", "--force-dist", "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 1
    assert json.loads(lines[0])["config"]["collective_ranks"] == 1


def _native_comm_worker(rank, world, port, out):
    """One rank over RCCL (all a one-GPU box can offer): the C-ABI collectives on their side stream."""
    import os
    import srl_amd
    from srl_amd import comm
    os.environ["SRL_COMM"] = "native"  # opt in: torch.distributed collectives are the default
    dev = f"cuda:{rank % torch.cuda.device_count()}"
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                            device_id=torch.device(dev))
    try:
        c = comm.NativeComm.from_process_group(dev)
        assert c is not None and c.world == world
        stats = torch.tensor([3.0, 1.5, 2.25], dtype=torch.float64, device=dev)
        grads = torch.arange(100000, dtype=torch.float32, device=dev)
        flat = torch.full((777,), 2.5 if rank == 0 else -1.0, dtype=torch.float32, device=dev)
        c.all_reduce_f64_async(stats)
        c.all_reduce_f32_async(grads[:50000])
        c.all_reduce_f32_async(grads[50000:])
        c.broadcast_async(flat, root=0)
        c.join()
        torch.cuda.synchronize()
        out[rank] = (stats.tolist(), float(grads.sum()), float(flat.sum()))
        c.close()
    finally:
        dist.destroy_process_group()


def test_native_rccl_wrappers_one_rank():
    """srl_comm_* / srl_allreduce_stats_f64x3 / srl_allreduce_grads / srl_broadcast_params through a real RCCL communicator
    bootstrapped from the process group (world size 1 leaves the data unchanged)."""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_native_comm_worker, args=(1, _free_port(), out), nprocs=1, join=True)
        stats, gsum, fsum = out[0]
    assert stats == [3.0, 1.5, 2.25]
    assert gsum == float(np.arange(100000, dtype=np.float32).astype(np.float64).sum()) or abs(gsum - 4999950000.0) < 1e4
    assert fsum == 777 * 2.5


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: RCCL wants a device per rank")
def test_native_rccl_wrappers_two_ranks():
    """The same wrappers with two real ranks (only where the box has two GPUs): sums double, rank 0's buffer wins the
    broadcast."""
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_native_comm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        for r in range(2):
            stats, gsum, fsum = out[r]
            assert stats == [6.0, 3.0, 4.5]
            assert abs(gsum - 2 * 4999950000.0) < 2e4 and fsum == 777 * 2.5
