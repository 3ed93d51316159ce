"""HBM observation ring (runtime/obs_ring.py): the frames a rollout uploaded are the frames the trainer reads.

CPU part: allocation / liveness / lease arithmetic and the ``RingObs`` view algebra.  GPU part: what the first layer reads
through the ring (staged frames, LayerNorm statistics) is BIT-IDENTICAL to what it reads from the host sample, whatever
share of the sample's stamps is still alive (all, some, none, or a ring too small to serve the sample at all -- the
plain-copy path); the loss statistics of the step are bit-identical too, the updated parameters agree to the run-to-run
noise of the float atomics in the backward pass (two host-fed steps differ by as much)."""
import numpy as np
import pytest
import torch

import srl_amd
from srl_amd.api import config, policy as policy_api, trainer as trainer_api
from srl_amd.namedarray import NamedArray
from srl_amd.runtime import synthetic
from srl_amd.runtime.ingest import SampleRing
from srl_amd.runtime.obs_ring import ObsLease, ObsRing, RingObs

srl_amd.register_all()

CNN_POLICY = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
                  popart=False, layernorm=False, shared_backbone=True, chunk_len=4, seed=5,
                  cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))
ATARI_TRAINER = dict(discount_rate=0.99, gae_lambda=0.97, eps_clip=0.2, clip_value=True, dual_clip=False,
                     value_loss='huber', value_loss_weight=1.0, value_loss_config=dict(delta=10.0),
                     entropy_bonus_weight=0.01, optimizer='adam', optimizer_config=dict(lr=5e-4), popart=False,
                     max_grad_norm=40.0, bootstrap_steps=1)


# ------------------------------------------------------------------------------------------------ CPU: bookkeeping
def cpu_ring(cap):
    return ObsRing({"obs": ("raw",)}, {"obs": (4,)}, cap, "cpu")


def test_allocation_never_straddles_the_end_and_liveness_follows_the_head():
    r = cpu_ring(10)
    assert r._alloc(4) == 0 and r._alloc(4) == 4
    assert r._alloc(4) == 10  # slots 8, 9 are skipped: the run starts a new lap at slot 0
    base = (r.generation << r.GEN_SHIFT) + 1  # stamp of sequence number 0
    refs = np.array([-1, 0, 3, 4, 7, 8, 10, 13, 14]) + base
    #        head = 14: rows 0..3 were lapped by 10..13, rows 4..7 are alive, 8 / 9 were never written but are in range
    assert r.alive(refs).tolist() == [False, False, False, True, True, True, True, True, False]
    # 0 (a zero-filled field), negatives and bare sequence numbers (no generation) are never alive
    assert r.alive(np.array([0, -1, 4, 7])).tolist() == [False] * 4
    with pytest.raises(BufferError):
        r._alloc(11)


def test_a_lease_stops_allocations_that_would_lap_it():
    r = cpu_ring(8)
    r._alloc(8)
    lease = ObsLease(r, 4)  # a bound sample whose oldest row is sequence 4
    r._leases.append(lease)
    assert r._alloc(4) == 8  # overwrites 0..3: below the lease
    with pytest.raises(BufferError):
        r._alloc(1)  # would overwrite sequence 4
    r.release(lease, record=False)
    assert r._alloc(1) == 12


def test_ringobs_view_algebra():
    r = cpu_ring(64)
    r.dtype["obs"] = torch.float32
    idx = torch.arange(30, dtype=torch.int32).reshape(5, 6)
    v = RingObs(r, "obs", (5, 6), index=idx)
    assert v.shape == (5, 6, 4) and v.rows == 30
    w = v[1:4]
    assert w.shape == (3, 6, 4) and w.index.tolist() == idx[1:4].tolist()
    f = w.reshape(18, 4)
    assert f.shape == (18, 4) and f.index.tolist() == idx[1:4].reshape(-1).tolist()
    assert f[6:12].index.reshape(-1).tolist() == idx[2].tolist()
    g = RingObs(r, "obs", (5, 2, 3), index=idx.reshape(5, 2, 3)).reshape(5, 6, 4)  # agents folded into the batch axis
    assert g.index.reshape(5, 6).tolist() == idx.tolist()
    s = RingObs(r, "obs", (12,), span=20)  # a run of slots (one rollout batch)
    assert s[4:8].span == 24 and s[4:8].rows == 4
    c = RingObs.cat([v[0:1], v[3:5]])
    assert c.shape == (3, 6, 4) and c.index.tolist() == torch.cat([idx[0:1], idx[3:5]]).tolist()
    with pytest.raises(ValueError):
        v.reshape(30, 5)
    with pytest.raises(TypeError):
        v[::2]


# ------------------------------------------------------------------------------------------------ GPU: the path
def _rollout_through_ring(infer, frames):
    """What the policy worker does over one rollout: every time row of the sample is an inference batch.  Returns the
    stamps [Tb, B, 1] the responses carried."""
    Tb, B = frames.shape[:2]
    refs = np.empty((Tb, B, 1), np.int64)
    for t in range(Tb):
        req = policy_api.RolloutRequest(obs=NamedArray(obs=frames[t]), is_evaluation=np.zeros((B, 1), np.uint8),
                                        on_reset=np.zeros((B, 1), np.uint8), client_id=np.zeros((B, 1), np.int32),
                                        request_id=np.arange(B).reshape(B, 1), received_time=np.zeros((B, 1), np.int64),
                                        buffer_index=np.zeros((B, 1), np.int32))
        res = infer.rollout(req)
        assert res.analyzed_result.obs_ref.shape == (B, 1) and res.analyzed_result.obs_ref.dtype == np.int64
        refs[t] = res.analyzed_result.obs_ref
    return refs


def _make(seed=5):
    return trainer_api.make(config.Trainer("mappo", args=dict(ATARI_TRAINER)),
                            config.Policy("actor-critic", args=dict(CNN_POLICY, seed=seed)))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["all-alive", "some-lapped", "no-stamps", "ring-too-small"])
def test_ring_fed_step_equals_the_host_fed_step(case):
    T, B = 12, 8
    Tb = T + 1
    arrays = synthetic.make_sample_arrays(seed=21, T=T, B=B, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=0.1)
    host_trainer, ring_trainer = _make(), _make()
    infer = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=99)))
    capacity = {"all-alive": 4 * Tb * B, "some-lapped": Tb * B + 2 * B, "no-stamps": 4 * Tb * B, "ring-too-small": 3 * B}[case]
    oring = infer.make_obs_ring(capacity, patch_rows=Tb * B if case == "no-stamps" else 3 * B)
    assert oring.layout == {"obs": ("s2d", 4)}
    infer.attach_obs_ring(oring)
    refs = _rollout_through_ring(infer, arrays["obs.obs"])
    if case == "some-lapped":  # later inference traffic laps the oldest rows of the sample before the trainer binds it
        _rollout_through_ring(infer, arrays["obs.obs"][:4])
    if case == "no-stamps":
        refs[:] = -1
    arrays = dict(arrays)
    arrays["analyzed_result.obs_ref"] = refs
    for step in range(2):
        host_sample = synthetic.to_sample_batch({k: v for k, v in arrays.items() if k != "analyzed_result.obs_ref"})
        want = host_trainer.step(host_sample)
        sample = synthetic.to_sample_batch(arrays)
        sring = SampleRing(sample[:, 0], batch_size=B, slots=1, device="cuda:0", obs_ring=oring)
        sring.put_batch(sample)
        before = dict(oring.stats)
        fed = sring.get_device()
        bound = isinstance(fed.obs.obs, RingObs)
        patched = oring.stats["rows_patched"] - before["rows_patched"]
        if case == "all-alive":
            assert bound and patched == 0
        elif case == "some-lapped":  # the two oldest time rows were lapped: uploaded into the patch area, every time
            assert bound and patched == 2 * B
        elif case == "no-stamps":
            assert bound and patched == Tb * B
        else:  # neither the ring nor its patch area can hold the sample: plain copies, as without a ring
            assert not bound and isinstance(fed.obs.obs, torch.Tensor) and oring.stats["binds_failed"] == step + 1
        if bound:  # the first layer's inputs, bit for bit: staged frames and statistics of every row of the sample
            from srl_amd import hip
            from srl_amd.algorithm.hipnet import Workspace
            n = Tb * B
            frames, mean, rstd = fed.obs.obs.reshape(n, 4, 84, 84).resolve(Workspace("cuda:0"), "check")
            raw = torch.from_numpy(arrays["obs.obs"]).to("cuda:0").reshape(n, 4, 84, 84)
            ref, rmean, rrstd = torch.empty_like(frames), torch.empty(n, device="cuda:0"), torch.empty(n, device="cuda:0")
            hip.obs_space_to_depth(raw.data_ptr(), True, n, 4, 84, 84, 4, ref.data_ptr(), rmean.data_ptr(), rrstd.data_ptr())
            assert torch.equal(frames, ref) and torch.equal(mean[:n], rmean) and torch.equal(rstd[:n], rrstd)
        got = ring_trainer.step(fed)
        sring.release(fed)
        for k in ("policy_loss", "value_loss", "entropy", "clip_ratio", "importance_weight", "advantage", "value_targets"):
            if step == 0:  # same parameters on both sides: the forward pass is bit-identical
                assert got.stats[k] == want.stats[k], (step, k, got.stats[k], want.stats[k])
            else:  # the parameters already differ by the atomics' noise of step 0
                assert abs(got.stats[k] - want.stats[k]) <= 1e-5 * max(abs(want.stats[k]), 1e-2), (step, k)
        assert abs(got.stats["grad_norm"] - want.stats["grad_norm"]) <= 1e-5 * want.stats["grad_norm"]
        # the backward pass accumulates with float atomics: two runs of the SAME feed differ in the last bits as well
        assert torch.allclose(host_trainer.policy.net.flat, ring_trainer.policy.net.flat, rtol=0, atol=2e-6), step
        assert np.array_equal(fed.analyzed_result.ret.cpu().numpy(), host_sample.analyzed_result.ret)


@pytest.mark.gpu
def test_rollout_reads_its_rows_from_the_ring_and_matches_the_plain_rollout():
    """The inference forward runs on the ring's rows (space-to-depth + statistics written there, not to a scratch buffer):
    same actions / log-probabilities / values as a policy without a ring, Philox stream included."""
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, size=(3, 40, 4, 84, 84), dtype=np.uint8)
    a = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=7)))
    b = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=7)))
    ring = b.make_obs_ring(100)  # 40-row batches: the third one starts a new lap (slots 80..99 skipped)
    b.attach_obs_ring(ring)
    seen = []
    for t in range(3):
        req = lambda: policy_api.RolloutRequest(obs=NamedArray(obs=frames[t]), is_evaluation=np.zeros((40, 1), np.uint8),
                                                on_reset=np.zeros((40, 1), np.uint8))
        ra, rb = a.rollout(req()), b.rollout(req())
        assert np.array_equal(ra.action.x, rb.action.x)
        assert np.array_equal(ra.analyzed_result.log_probs, rb.analyzed_result.log_probs)
        assert np.array_equal(ra.analyzed_result.value, rb.analyzed_result.value)
        assert "obs_ref" not in list(ra.analyzed_result.keys())
        seen.append(rb.analyzed_result.obs_ref[:, 0])
    base = (ring.generation << ring.GEN_SHIFT) + 1  # a stamp = generation | sequence number + 1: never 0
    assert seen[0].tolist() == list(range(base, base + 40)) and seen[1].tolist() == list(range(base + 40, base + 80))
    assert seen[2].tolist() == list(range(base + 100, base + 140))
    assert ring.alive(np.concatenate(seen)).tolist() == [False] * 40 + [True] * 80
    # what a step that never went through inference carries (a zero-filled analyzed_result), negatives, and the stamps of
    # ANOTHER ring (a recreated one, another policy worker's) are dead here whatever this ring's head is
    other = b.make_obs_ring(100)
    assert other.generation != ring.generation
    foreign = seen[2] - base + ((other.generation << other.GEN_SHIFT) + 1)
    assert not ring.alive(np.array([0, -1, -7], dtype=np.int64)).any() and not ring.alive(foreign).any()
    assert not other.alive(seen[2]).any()


def _stacked_episode_frames(rng, T, B, p_reset=0.15):
    """Frame stacks as the reference's `FrameStack` wrapper produces them (atari_wrappers.py:211-242): obs[t, b] = the four
    latest planes of environment b, newest last; at a reset the stack is four copies of the first plane.  Returns
    (stacks [T, B, 4, 84, 84], newest planes [T, B, 1, 84, 84], reset flags [T, B])."""
    planes = rng.integers(0, 256, size=(T, B, 1, 84, 84), dtype=np.uint8)
    reset = rng.random((T, B)) < p_reset
    reset[0] = True
    stacks = np.empty((T, B, 4, 84, 84), np.uint8)
    for t in range(T):
        for b in range(B):
            if reset[t, b]:
                stacks[t, b] = planes[t, b]
            else:
                stacks[t, b, :3] = stacks[t - 1, b, 1:]
                stacks[t, b, 3] = planes[t, b, 0]
    return stacks, planes, reset


@pytest.mark.gpu
def test_stack_aware_requests_match_whole_stack_requests_bit_for_bit():
    """Requests that carry the newest plane + the stamp of the previous observation (`ring_prev`; 0 at an episode start): the
    ring assembles the rows (srl_ring_stack_push).  Same staged bytes and LayerNorm statistics as whole-stack requests, same
    actions / log-probabilities / values (Philox stream included), stamps that bind the same rows for the trainer; a previous
    observation the ring has lapped raises LookupError."""
    rng = np.random.default_rng(11)
    T, B = 7, 24
    stacks, planes, reset = _stacked_episode_frames(rng, T, B)
    a = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=7)))
    b = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=7)))
    ring_a, ring_b = a.make_obs_ring(T * B), b.make_obs_ring(T * B)
    a.attach_obs_ring(ring_a)
    b.attach_obs_ring(ring_b)
    prev = np.zeros((B, 1), np.int64)
    refs_a, refs_b = [], []
    for t in range(T):
        flags = dict(is_evaluation=np.zeros((B, 1), np.uint8), on_reset=reset[t].astype(np.uint8).reshape(B, 1))
        ra = a.rollout(policy_api.RolloutRequest(obs=NamedArray(obs=stacks[t]), **flags))
        prev[reset[t], 0] = 0   # the actor: no predecessor at an episode start
        rb = b.rollout(policy_api.RolloutRequest(obs=NamedArray(obs=planes[t], ring_prev=prev.copy()), **flags))
        assert np.array_equal(ra.action.x, rb.action.x)
        assert np.array_equal(ra.analyzed_result.log_probs, rb.analyzed_result.log_probs)
        assert np.array_equal(ra.analyzed_result.value, rb.analyzed_result.value)
        prev = rb.analyzed_result.obs_ref.copy()   # the stamp the actor keeps for this environment's next request
        refs_a.append(ra.analyzed_result.obs_ref)
        refs_b.append(rb.analyzed_result.obs_ref)
    assert torch.equal(ring_a.storage["obs"][:T * B], ring_b.storage["obs"][:T * B])
    assert torch.equal(ring_a.mean["obs"][:T * B], ring_b.mean["obs"][:T * B]) and torch.equal(ring_a.rstd["obs"][:T * B], ring_b.rstd["obs"][:T * B])
    assert ring_b.stats["rows_put_stacked"] == T * B and ring_b.stats["rows_put"] == T * B
    # the trainer's side: the stamps bind the same rows
    bound_a, lease_a = ring_a.bind(np.stack(refs_a))
    bound_b, lease_b = ring_b.bind(np.stack(refs_b))
    assert torch.equal(bound_a["obs"].index, bound_b["obs"].index)
    ring_a.release(lease_a)
    ring_b.release(lease_b)
    # a predecessor the ring no longer holds: the ring says so and the caller sends the whole stack
    for _ in range(2):   # two more laps' worth of rows
        ring_b.put({"obs": torch.from_numpy(stacks[:4].reshape(-1, 4, 84, 84)).cuda()})
    from srl_amd.runtime.obs_ring import WholeStackNeeded
    head = ring_b.head
    with pytest.raises(WholeStackNeeded):   # (a LookupError: the policy's defined per-request refusal)
        b.rollout(policy_api.RolloutRequest(obs=NamedArray(obs=planes[0], ring_prev=refs_b[0]), is_evaluation=np.zeros((B, 1), np.uint8),
                                            on_reset=np.zeros((B, 1), np.uint8)))
    foreign = refs_a[-1]   # another ring's stamps are not predecessors here either
    with pytest.raises(LookupError):
        b.rollout(policy_api.RolloutRequest(obs=NamedArray(obs=planes[0], ring_prev=foreign), is_evaluation=np.zeros((B, 1), np.uint8),
                                            on_reset=np.zeros((B, 1), np.uint8)))
    unstaged = np.full((B, 1), -1, np.int64)   # what `put_or_skip` hands out for a row it could not stage
    with pytest.raises(WholeStackNeeded):
        b.rollout(policy_api.RolloutRequest(obs=NamedArray(obs=planes[0], ring_prev=unstaged), is_evaluation=np.zeros((B, 1), np.uint8),
                                            on_reset=np.zeros((B, 1), np.uint8)))
    assert ring_b.head == head   # a refused request consumed no ring slots (and so lapped nobody else's predecessor)


@pytest.mark.gpu
def test_zero_stamps_are_patched_and_a_full_ring_degrades_instead_of_failing():
    """(a) Rows whose stamp is 0 while the ring's head is still below its capacity -- terminal observations that never went
    through inference: the reference actor zero-fills their analyzed_result -- are uploaded from the host copy, not bound to
    sequence number 0's frame.  (b) A rollout that would lap rows a training step has leased stages nothing and still answers;
    its rows carry dead stamps."""
    rng = np.random.default_rng(11)
    pol = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=3)))
    Tb, B = 3, 8
    oring = pol.make_obs_ring(64, patch_rows=Tb * B)
    pol.attach_obs_ring(oring)
    frames = rng.integers(0, 256, size=(Tb, B, 4, 84, 84), dtype=np.uint8)
    refs = _rollout_through_ring(pol, frames)
    refs[1, 2:5] = 0  # never sent for inference
    host = {"obs": torch.from_numpy(frames)}
    bound = oring.bind(refs, host)
    assert bound is not None and oring.stats["rows_patched"] == 3
    rows, lease = bound
    assert lease.min_seq == 0
    from srl_amd import hip
    from srl_amd.algorithm.hipnet import Workspace
    n = Tb * B
    got, mean, rstd = rows["obs"].reshape(n, 4, 84, 84).resolve(Workspace("cuda:0"), "check")
    raw = torch.from_numpy(frames).to("cuda:0").reshape(n, 4, 84, 84)
    ref, rm, rr = torch.empty_like(got), torch.empty(n, device="cuda:0"), torch.empty(n, device="cuda:0")
    hip.obs_space_to_depth(raw.data_ptr(), True, n, 4, 84, 84, 4, ref.data_ptr(), rm.data_ptr(), rr.data_ptr())
    assert torch.equal(got, ref) and torch.equal(mean[:n], rm)
    # (b) the lease pins sequence 0 on: 64 - 24 = 40 free rows; a 48-row request cannot be staged
    big = rng.integers(0, 256, size=(48, 4, 84, 84), dtype=np.uint8)
    plain = policy_api.make(config.Policy("actor-critic", args=dict(CNN_POLICY, seed=3)))
    req = lambda: policy_api.RolloutRequest(obs=NamedArray(obs=big), is_evaluation=np.ones((48, 1), np.uint8),
                                            on_reset=np.zeros((48, 1), np.uint8))
    ra, rb = plain.rollout(req()), pol.rollout(req())
    assert oring.stats["puts_unstaged"] == 1
    assert np.array_equal(ra.action.x, rb.action.x) and np.array_equal(ra.analyzed_result.value, rb.analyzed_result.value)
    assert (rb.analyzed_result.obs_ref == -1).all() and not oring.alive(rb.analyzed_result.obs_ref[:, 0]).any()
    oring.release(lease)
    rc = pol.rollout(req())  # released: staged again
    assert oring.alive(rc.analyzed_result.obs_ref[:, 0]).all() and oring.stats["puts_unstaged"] == 1


@pytest.mark.gpu
def test_vector_observations_go_through_the_ring_as_raw_rows():
    pol = dict(obs_dim=4, action_dim=2, hidden_dim=32, num_dense_layers=1, num_rnn_layers=0, popart=False, layernorm=True,
               shared_backbone=False, seed=4)
    T, B = 16, 6
    arrays = synthetic.make_sample_arrays(seed=5, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.1)
    mk = lambda: trainer_api.make(config.Trainer("mappo", args=dict(popart=False)), config.Policy("actor-critic", args=pol))
    host_trainer, ring_trainer = mk(), mk()
    infer = policy_api.make(config.Policy("actor-critic", args=pol))
    oring = infer.make_obs_ring(1000)
    assert oring.layout == {"obs": ("raw",)}
    infer.attach_obs_ring(oring)
    refs = _rollout_through_ring(infer, arrays["obs.obs"])
    sample = synthetic.to_sample_batch(dict(arrays, **{"analyzed_result.obs_ref": refs}))
    sring = SampleRing(sample[:, 0], batch_size=B, slots=1, device="cuda:0", obs_ring=oring)
    sring.put_batch(sample)
    fed = sring.get_device()
    assert isinstance(fed.obs.obs, RingObs) and oring.stats["rows_patched"] == 0
    got = ring_trainer.step(fed)
    sring.release(fed)
    want = host_trainer.step(synthetic.to_sample_batch(arrays))
    for k in ("policy_loss", "value_loss", "entropy", "clip_ratio", "importance_weight", "advantage", "value_targets"):
        assert got.stats[k] == want.stats[k], k
    assert torch.allclose(host_trainer.policy.net.flat, ring_trainer.policy.net.flat, rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_gather_rows_kernel():
    from srl_amd import hip
    g = torch.Generator(device="cuda:0").manual_seed(0)
    for row_bytes, n, cap in ((28224, 300, 500), (16, 1000, 64), (4, 777, 100), (1040, 5, 9)):
        src = torch.randint(0, 256, (cap, row_bytes), dtype=torch.uint8, device="cuda:0", generator=g)
        idx = torch.randint(0, cap, (n,), dtype=torch.int32, device="cuda:0", generator=g)
        dst = torch.empty((n, row_bytes), dtype=torch.uint8, device="cuda:0")
        hip.gather_rows(src.data_ptr(), row_bytes, idx, n, dst.data_ptr())
        assert torch.equal(dst, src[idx.long()]), row_bytes
