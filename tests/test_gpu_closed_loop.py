"""The actor -> policy worker -> buffer -> trainer -> parameter swap loop of the hot path, closed, in one process.

Real environments (CartPole) are stepped with actions SAMPLED by the HIP rollout path (Philox), their requests folded by
``InferenceBatcher`` (reference ``policy_worker.py:162-242``), trajectories cut into overlapping ``[Tb]`` samples the way
the actor worker does (``actor_worker.py:141-161,278-321``), written column by column into the
pinned ``SampleRing``, trained on by ``MultiAgentPPO.step``, and the new parameters handed back to the inference policy
through the batcher's ``parameter_source`` (``policy_worker.py:166-172``).  Nothing of the reference's worker runtime
(processes, streams, name service) is involved: this checks that the sampled path and the version plumbing work
TOGETHER -- the return must rise and every sample must carry the version of the parameters that produced it.
"""
import numpy as np
import pytest
import torch

import srl_amd
from srl_amd.algorithm.ppo_types import PPORolloutAnalyzedResult
from srl_amd.api import config, policy as policy_api, trainer as trainer_api
from srl_amd.api.env_utils import DiscreteAction
from srl_amd.api.trainer import SampleBatch
from srl_amd.envs.cartpole import CartPoleEnvironment
from srl_amd.namedarray import NamedArray, recursive_aggregate
from srl_amd.runtime.batcher import InferenceBatcher
from srl_amd.runtime.ingest import SampleRing
from srl_amd.runtime.obs_ring import RingObs

srl_amd.register_all()
pytestmark = pytest.mark.gpu

POLICY = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False, layernorm=True,
              shared_backbone=False, seed=3)
TRAINER = dict(popart=False, ppo_epochs=4, optimizer_config=dict(lr=2e-3), max_grad_norm=5.0, entropy_bonus_weight=0.005)
N_ENVS, T, BOOT = 16, 32, 1


class Actor:
    """One environment and the sample cutting of the reference's actor worker: a sample is Tb = T + bootstrap consecutive
    steps, consecutive samples overlap by the bootstrap rows (actor_worker.py:152-155)."""

    def __init__(self, idx):
        self.idx = idx
        self.env = CartPoleEnvironment(seed=1000 + idx)
        self.cur = self.env.reset()[0]
        self.on_reset = 1
        self.steps = []
        self.returns = []
        self.t = 0

    def request(self):
        self.t += 1
        return policy_api.RolloutRequest(obs=NamedArray(obs=self.cur.obs["obs"][None]), on_reset=np.array([[self.on_reset]], np.uint8),
                                         is_evaluation=np.zeros((1, 1), np.uint8), client_id=np.array([[self.idx]], np.int32),
                                         request_id=np.array([[self.t]], np.int32), received_time=np.zeros((1, 1), np.int64),
                                         buffer_index=np.full((1, 1), -1, np.int32), step_count=np.full((1, 1), self.t, np.int32),
                                         ready=np.ones((1, 1), np.bool_))

    def act(self, action, log_prob, value, version, obs_ref):
        """Record the step taken at the current observation and advance the environment (Appendix B row conventions)."""
        cur = self.cur
        done, trunc = int(cur.done[0]), int(0 if cur.truncated is None else cur.truncated[0])
        terminal = bool(done or trunc)
        if terminal:  # a terminal observation: no action is executed from it, the next observation opens a new episode
            self.returns.append(float(cur.info["episode_return"][0]))
            nxt, reward, next_on_reset = self.env.reset()[0], np.zeros(1, np.float32), 1
        else:
            nxt = self.env.step([DiscreteAction(np.asarray(action))])[0]
            reward, next_on_reset = nxt.reward.astype(np.float32), 0
        self.steps.append(SampleBatch(
            obs=NamedArray(obs=cur.obs["obs"]), on_reset=np.array([self.on_reset], np.uint8), done=np.array([done], np.uint8),
            truncated=np.array([trunc], np.uint8), action=DiscreteAction(np.asarray(action, np.int32).reshape(1)), reward=reward,
            info=NamedArray(episode_return=cur.info["episode_return"].astype(np.float32)),
            info_mask=np.array([terminal], np.uint8),
            # the response's analyzed_result is stored as it came (actor_worker.py:521-535): with an observation ring
            # attached to the inference policy it carries the ring stamp of this observation
            analyzed_result=PPORolloutAnalyzedResult(log_probs=np.asarray(log_prob, np.float32).reshape(1),
                                                     value=np.asarray(value, np.float32).reshape(1),
                                                     obs_ref=np.asarray(obs_ref, np.int64).reshape(1)),
            policy_version_steps=np.array([version], np.int64)))
        self.cur, self.on_reset = nxt, next_on_reset

    def pop_sample(self):
        """A [Tb, ...] trajectory once Tb steps have accumulated; its last BOOT rows open the next one."""
        if len(self.steps) < T + BOOT:
            return None
        traj = recursive_aggregate(self.steps[:T + BOOT], np.stack)
        self.steps = self.steps[T:]
        return traj


def test_cartpole_closed_loop_learns_and_versions_advance():
    trainer = trainer_api.make(config.Trainer("mappo", args=TRAINER), config.Policy("actor-critic", args=POLICY))
    # the inference replica is its own policy object: it only ever sees parameters through the checkpoint hand-over
    infer = policy_api.make(config.Policy("actor-critic", args=dict(POLICY, seed=99)))
    fresh = [trainer.get_checkpoint()]  # size-1 "queue" of parameters waiting to be swapped in (policy_worker.py:166-172)

    def parameter_source():
        return fresh.pop() if fresh else None

    # observations stay in HBM from the rollout that saw them to the update that trains on them (runtime/obs_ring.py)
    obs_ring = infer.make_obs_ring(4 * N_ENVS * (T + BOOT))
    infer.attach_obs_ring(obs_ring)
    batcher = InferenceBatcher(infer, policy_name="cartpole", batch_size=N_ENVS, parameter_source=parameter_source)
    actors = [Actor(i) for i in range(N_ENVS)]
    ring = None
    updates, curve, version_log = 0, [], []
    max_updates = 120
    while updates < max_updates:
        for a in actors:
            batcher.post(a.request())
        (resp,) = batcher.poll()  # all requests fold into one batch of N_ENVS rows
        assert resp.action.x.shape == (N_ENVS, 1) and resp.policy_version_steps.shape == (N_ENVS, 1)
        order = resp.client_id[:, 0]
        assert sorted(order.tolist()) == list(range(N_ENVS))
        for row, cid in enumerate(order):
            actors[cid].act(resp.action.x[row], resp.analyzed_result.log_probs[row], resp.analyzed_result.value[row],
                            int(resp.policy_version_steps[row, 0]), resp.analyzed_result.obs_ref[row])
        for a in actors:
            traj = a.pop_sample()
            if traj is None:
                continue
            if ring is None:
                ring = SampleRing(traj, batch_size=N_ENVS, slots=2, device="cuda:0", obs_ring=obs_ring)
            ring.put_column(traj)
        if ring is not None and ring.ready():
            batch = ring.get_device()
            assert isinstance(batch.obs.obs, RingObs)  # bound to the rows the rollouts left in HBM, not uploaded again
            stamps = batch.policy_version_steps.cpu().numpy()
            version_log.append((int(stamps.min()), int(stamps.max()), trainer.policy.version))
            res = trainer.step(batch)
            ring.release(batch)
            updates += 1
            assert res.step == trainer.policy.version == updates - 1  # the reference's versions start at -1 (api/policy.py)
            fresh[:] = [trainer.get_checkpoint()]  # newest parameters replace any not yet taken
            recent = [r for a in actors for r in a.returns[-4:]]
            curve.append(float(np.mean(recent)) if recent else 0.0)
            if updates >= 30 and curve[-1] >= 150.0:
                break
    # the policy improved: random play lasts ~20-25 steps
    early, late = float(np.mean(curve[2:8])), float(np.max(curve[-10:]))
    assert early < 60.0, curve[:10]
    assert late >= 100.0 and late >= 3.0 * early, (early, late, curve[::5])
    # version plumbing: a batch holds samples produced under the parameters of the last one or two updates -- never newer
    # than the trainer, and the stamps advance with the updates
    for lo, hi, ver in version_log:
        assert -1 <= lo <= hi <= ver and ver - lo <= 2, (lo, hi, ver)
    assert version_log[-1][1] >= version_log[5][1] + (len(version_log) - 6) - 1
    assert infer.version == trainer.policy.version - 1 or infer.version == trainer.policy.version
    # the inference replica really runs the trainer's parameters (handed over by checkpoint)
    batcher.post(actors[0].request())
    batcher.poll()
    assert infer.version == trainer.policy.version
    assert torch.equal(infer.net.flat, trainer.policy.net.flat)
    assert np.isfinite(list(res.stats.values())).all() and res.stats.get("episode_return", 1.0) > 0
    assert obs_ring.stats["rows_patched"] == 0 and obs_ring.stats["binds"] == updates and obs_ring.stats["binds_failed"] == 0
    print(f"closed loop: {updates} updates, mean return {early:.1f} -> {late:.1f}; curve {[round(c) for c in curve[::8]]}")
